"""Host side of a frame alone (no GPU needed: the SIMT-emulated library has the same host code): lister threads -> chunk
preparation -> staged uploads, timed per frame; frame_end is never called.  usage: lister_bench.py [w h threads frames tile_cols tile_rows]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import util
from dav1d_amd import api
import e2e
import synth_lib

a = [int(v) for v in sys.argv[1:]]
w, h, threads, frames, tcols, trows = (a + [3840, 2160, 8, 5, 4, 2][len(a):])[:6]
ctx = util.make_ctx() if hasattr(util, "make_ctx") else api.Context(0, lib_path=util.emu_lib_path())
layout, bpc = api.LAYOUT_I420, 10
ho = e2e.HandOff(w, h, layout, bpc, True, tcols, trows)
sp = e2e.c2_params(0xE2E)
assert synth_lib.synth_frame(ho.desc, sp, ho.cf.ctypes.data, ho.cf.nbytes, len(ho.cbi), None, 0) == 0
refs = [ctx.picture(w, h, layout, bpc) for _ in range(3)]
refs7 = [refs[i % 3] for i in range(7)]
cur = ctx.picture(w, h, layout, bpc)
n_tcols, n_trows = ho.desc.n_tile_cols, ho.desc.n_tile_rows
ts = []
with ThreadPoolExecutor(threads) as ex:
    for it in range(frames):
        t0 = time.perf_counter()
        frame = ctx.frame(cur, refs7)
        lh = C.c_void_p()
        assert ctx.lib.dav1d_hip_lister_create(C.byref(lh), C.byref(ho.desc), frame.h) == 0

        def tile(k):
            tr, tc = divmod(k, n_tcols)
            for sby in range(ho.rows[tr], ho.rows[tr + 1]):
                assert ctx.lib.dav1d_hip_lister_tile_sbrow(lh, tr, tc, sby) == 0
        if os.environ.get("LISTER_NATIVE"):
            assert ctx.lib.dav1d_hip_lister_run(lh, threads) == 0
        else:
            list(ex.map(tile, range(n_tcols * n_trows)))
        ts.append((time.perf_counter() - t0) * 1e3)
        ctx.lib.dav1d_hip_lister_destroy(lh)
        frame.destroy()
print("%dx%d, %d threads over %d x %d tiles: list_ms per frame %s  (median of the last %d: %.2f)" %
      (w, h, threads, n_tcols, n_trows, [round(t, 1) for t in ts], len(ts) - 1, float(np.median(ts[1:]))))
