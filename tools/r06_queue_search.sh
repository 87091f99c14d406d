#!/bin/bash
# the step over pair streams x stream padding (per context), two frame contexts, 60 steps
mkdir -p gpurun_out/r06g
run() { local name="$1"; shift
  env "$@" python bench.py --step-only --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'])"; }
for ps in 2 1 3; do
  for p in 0 1 2 3 "0,1" "0,2" "0,3" "1,0" "1,2" "1,3" "2,1" "2,3" "3,1" "1,5" "5,1"; do
    run "ps $ps pad $p" DAV1D_HIP_RECON_PAIR_STREAMS=$ps DAV1D_HIP_STREAM_PAD=$p
  done
done 2>&1 | tee gpurun_out/r06g/queue_search.txt
for prio in 1 -1; do
  for ps in 2 3; do
    for p in 0 1 2 "0,2"; do
      run "prio $prio ps $ps pad $p" DAV1D_HIP_PAIR_PRIORITY=$prio DAV1D_HIP_RECON_PAIR_STREAMS=$ps DAV1D_HIP_STREAM_PAD=$p
    done
  done
done 2>&1 | tee -a gpurun_out/r06g/queue_search.txt
sort -k6 -n gpurun_out/r06g/queue_search.txt | head -12
