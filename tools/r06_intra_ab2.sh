#!/bin/bash
# the inter frame's intra blocks: superblock route (intra_sb = 2, default) against the stepped launches (intra_sb = 1: only long wavefronts go superblock by superblock)
mkdir -p gpurun_out/r06
for v in "sb2:DAV1D_HIP_INTRA_SB=2" "sb1:DAV1D_HIP_INTRA_SB=1" "sb0:DAV1D_HIP_INTRA_SB=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e DAV1D_HIP_TRACE_INTRA=1 python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-c1 --no-pmc --no-inflight > gpurun_out/r06/full2_$n.json 2> gpurun_out/r06/full2_$n.err
  grep -m3 "intra" gpurun_out/r06/full2_$n.err | cut -c1-300
  python - "$n" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r06/full2_%s.json'%n).read().strip().splitlines()[-1])
    print(n, json.dumps(d['legs']['full_table'])[:400])
except Exception as e: print(n, 'bench failed', e)
P
done
