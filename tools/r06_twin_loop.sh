#!/bin/bash
# dav1d's task loop (pass 1 injected) with the backend's pictures raster (ref_twin 1), retiled at frame end (2), tiled-native where a frame allows (3)
for t in 1 2 3 1 3; do
  DAV1D_HIP_REF_TWIN=$t python tools/hooked_probe.py --frames 24 --check-frames 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ref_twin $t', {k:d.get(k) for k in ('fps','ms_per_frame','parity','ms_per_frame_by_stage_summed_over_threads')})"
  DAV1D_HIP_REF_TWIN=$t python tools/hooked_probe.py --frames 24 --check-frames 2 --intra-pct 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ref_twin $t intra 0', {k:d.get(k) for k in ('fps','ms_per_frame','parity')})"
done
