#!/bin/bash
# which streams share a hardware queue: the paired launches on side streams 1.. / 2.. / 3.., unused streams in front of a context's side streams, per context ("a,b": first / second context)
mkdir -p gpurun_out/r06g
run() { # name, env...
  local name="$1"; shift
  env "$@" python bench.py --step-only --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'])"
}
for rep in 1 2; do
  run "default" X=1
  for f in 1 3 "2,1" "2,3" "1,2" "3,2" "1,3" "3,1"; do run "pair_first $f" DAV1D_HIP_RECON_PAIR_FIRST=$f; done
  for p in 1 2 3 "0,1" "0,2" "0,3" "1,0" "2,0" "3,0" "1,2" "2,1"; do run "pad $p" DAV1D_HIP_STREAM_PAD=$p; done
  for f in 1 2 3; do run "q8 pair_first $f" GPU_MAX_HW_QUEUES=8 DAV1D_HIP_RECON_PAIR_FIRST=$f; done
  for p in 1 2 3 "0,1" "0,2" "0,4"; do run "q8 pad $p" GPU_MAX_HW_QUEUES=8 DAV1D_HIP_STREAM_PAD=$p; done
done 2>&1 | tee gpurun_out/r06g/queue_map.txt
