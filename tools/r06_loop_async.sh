#!/bin/bash
# dav1d's task loop (8K, 4 tile columns, frame delay 8, 64 threads, pass 1 injected: bench.py's dav1d_task_loop leg) with the preparation on the listing threads (0) and on the library's (1)
for m in 0 1 0 1 0 1; do
  DAV1D_HIP_PREP_ASYNC=$m python tools/hooked_probe.py --frames 24 --check-frames 2 --delay 8 --threads 64 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('prep_async $m', 'steady', (d.get('steady_state') or {}).get('fps'), 'all', d.get('fps'), 'listing', (d.get('ms_per_frame_by_stage_summed_over_threads') or {}).get('listing'))"
done
