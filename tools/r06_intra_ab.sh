#!/bin/bash
# A/B of the one-wave superblock form (intra_sb_waves = 1; intra_sb_one_below): key frame / inter frame with 10 % intra / key frame with copies, and the full-table leg's stages
mkdir -p gpurun_out/r06
for v in "old:DAV1D_HIP_INTRA_SB_ONE_BELOW=0" "auto:DAV1D_HIP_INTRA_SB_ONE_BELOW=12" "one:DAV1D_HIP_INTRA_SB_WAVES=1" "auto30:DAV1D_HIP_INTRA_SB_ONE_BELOW=30"; do
  n=${v%%:*}; e=${v#*:}
  env $e python tools/intra_ab.py --check > gpurun_out/r06/intra_ab_$n.json 2> gpurun_out/r06/intra_ab_$n.err
  env $e python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-c1 --no-pmc --no-inflight > gpurun_out/r06/full_$n.json 2> gpurun_out/r06/full_$n.err
  python - "$n" <<'P'
import json,sys
n=sys.argv[1]
print(n, open('gpurun_out/r06/intra_ab_%s.json'%n).read().strip()[-400:])
try:
    d=json.loads(open('gpurun_out/r06/full_%s.json'%n).read().strip().splitlines()[-1])
    print(n, json.dumps(d['legs']['full_table'])[:400])
except Exception as e: print(n, 'bench failed', e)
P
done
