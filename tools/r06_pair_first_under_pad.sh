#!/bin/bash
# the first side stream of the paired launches (DAV1D_HIP_RECON_PAIR_FIRST) under the default padding, two frame contexts, 60 and 20 steps
mkdir -p gpurun_out/r06g
run() { local name="$1"; shift; local st="$1"; shift
  env "$@" python bench.py --step-only --steps $st --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name steps $st', d['ms_per_step'])"; }
for rep in 1 2; do
  for f in 2 1 3 "2,3" "3,2" "1,2" "2,1" "1,3" "3,1"; do
    run "pair_first $f" 60 DAV1D_HIP_RECON_PAIR_FIRST=$f
  done
done 2>&1 | tee gpurun_out/r06g/pair_first_under_pad.txt
