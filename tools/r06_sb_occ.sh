#!/bin/bash
# the superblock kernel asked for 3 / 4 waves per SIMD (168 / 128 registers, spills) against 2 (256): key frame, inter frame with 10 % intra, key frame with copies
for v in "" sb3 sb4 ""; do
  lib=""; [ -n "$v" ] && lib=dav1d_amd/build/variants/$v.so
  python tools/intra_ab.py $lib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-tree}', {k:(x['frame_end_ms'] if isinstance(x,dict) else x) for k,x in d.items()})"
  DAV1D_HIP_INTRA_SB_WAVES=4 python tools/intra_ab.py $lib 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-tree} 4 waves forced', {k:(x['frame_end_ms'] if isinstance(x,dict) else x) for k,x in d.items()})"
done
