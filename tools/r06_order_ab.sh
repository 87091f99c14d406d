#!/bin/bash
# the paired blocks with two predictions first (DAV1D_HIP_PAIR_LONG_FIRST=1) against spatial order: per-kernel times and the step with two frames in flight
for v in 0 1 0 1; do
  DAV1D_HIP_PAIR_LONG_FIRST=$v python tools/layout_sweep.py --no-raster --kernels --sets "" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('long_first $v one frame', d['ms_per_step'], d['kernels_us'])"
  DAV1D_HIP_PAIR_LONG_FIRST=$v python tools/layout_sweep.py --no-raster --steps 60 --inflight 2 --sets "" "" 2>/dev/null | tail -2 | python -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print('long_first $v two frames', d['ms_per_step'])"
done
