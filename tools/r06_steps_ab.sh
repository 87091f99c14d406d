#!/bin/bash
# the driver's command times 20 steps behind 5 warm-up steps: what does a run of 20 lose against one of 60, and is it the clocks
# (the arena fill — 14 GB of copy kernels — directly in front of the warm-up instead of the host-to-device measurement's DMA)?
mkdir -p gpurun_out/r06g
for rep in 1 2; do
  for cfg in "0 20 5" "1 20 5" "0 20 40" "1 20 40" "0 60 5" "1 60 5"; do
    set -- $cfg
    BENCH_H2D_FIRST=$1 python bench.py --step-only --steps $2 --warmup $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('h2d_first $1 steps $2 warmup $3', d['ms_per_step'], d['value'])"
  done
done 2>&1 | tee gpurun_out/r06g/steps_ab.txt
# hardware queues: the step runs on 2 frame contexts x (main + 2 side streams + ...) streams; the runtime maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
for rep in 1 2; do
  for q in 4 8 16; do
    GPU_MAX_HW_QUEUES=$q python bench.py --step-only --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw_queues $q steps 60', d['ms_per_step'], d['value'])"
    [ $rep = 1 ] && GPU_MAX_HW_QUEUES=$q python bench.py --step-only --steps 60 --warmup 5 --frame-contexts 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw_queues $q steps 60 fc 3', d['ms_per_step'], d['value'])"
  done
done 2>&1 | tee gpurun_out/r06g/hw_queues.txt
