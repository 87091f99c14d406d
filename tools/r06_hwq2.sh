#!/bin/bash
mkdir -p gpurun_out/r06g
for rep in 1 2 3; do
  for q in 4 5 6 7; do
    GPU_MAX_HW_QUEUES=$q python bench.py --step-only --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw_queues $q steps 60', d['ms_per_step'], d['value'])"
  done
done 2>&1 | tee gpurun_out/r06g/hw_queues3.txt
for q in 5 6; do
GPU_MAX_HW_QUEUES=$q python bench.py --step-only --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw_queues $q steps 20', d['ms_per_step'], d['value'])"
GPU_MAX_HW_QUEUES=$q python bench.py --step-only --steps 60 --warmup 5 --frame-contexts 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hw_queues $q steps 60 fc 3', d['ms_per_step'], d['value'])"
done 2>&1 | tee -a gpurun_out/r06g/hw_queues3.txt
