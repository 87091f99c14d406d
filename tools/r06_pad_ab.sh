#!/bin/bash
# unused streams in front of a context's side streams (DAV1D_HIP_STREAM_PAD): the other shapes of the step
mkdir -p gpurun_out/r06g
run() { local name="$1"; shift; local e="$1"; shift
  env $e python bench.py --step-only "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'])"; }
for rep in 1 2; do
for p in 0 1 "0,2"; do
  run "8K fc2 20 steps pad $p" DAV1D_HIP_STREAM_PAD=$p --steps 20 --warmup 5
  run "8K fc2 60 steps pad $p" DAV1D_HIP_STREAM_PAD=$p --steps 60 --warmup 5
  run "8K fc1 pad $p" DAV1D_HIP_STREAM_PAD=$p --steps 60 --warmup 5 --frame-contexts 1
  run "8K fc3 pad $p" DAV1D_HIP_STREAM_PAD=$p --steps 60 --warmup 5 --frame-contexts 3
  run "4K c1 fc2 pad $p" DAV1D_HIP_STREAM_PAD=$p --steps 60 --warmup 5 --width 3840 --height 2160 --bpc 8 --mix c1
  run "4K c1 fc1 pad $p" DAV1D_HIP_STREAM_PAD=$p --steps 60 --warmup 5 --width 3840 --height 2160 --bpc 8 --mix c1 --frame-contexts 1
done
done 2>&1 | tee gpurun_out/r06g/pad_ab.txt
