"""End-to-end route with different lister thread pools / upload modes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dav1d_amd import api, e2e
ctx = api.Context(0)
for up in (1, 0):
    ctx.set_option("chunk_upload", up)
    for native, thr in ((False, 32), (True, 32), (True, 64), (True, 105)):
        r = e2e.run(ctx, frames=5, threads=thr, tile_cols=16, tile_rows=8, native_threads=native)
        print("per-chunk upload" if up else "one upload      ", "native" if native else "python", thr, {k: r[k] for k in ("list_ms", "frame_end_ms", "total_ms", "value")})
