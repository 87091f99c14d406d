"""Which family of the reference-signature DSP table makes the reference's own pass 2 + filters differ (debug aid for
lister_util.c0_line): the reference's table with ONE member range replaced by the HIP library's at a time."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util, lister_util as lu
w, h, bpc, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1920, int(sys.argv[2]) if len(sys.argv) > 2 else 1080, 8, 0xC0
ctx = util.make_context(os.environ.get("BACKEND", "hip")); ctx.backend = "hip"
filters = dict(lf=(20, 28, 16, 24, 0, False), cdef=(5, 2, [17, 33, 0, 63], [5, 0, 20, 48]), lr=([1, 1, 1], [6, 6]))
RANGES = {"ipred": (8, 32), "mc_put": (32, 42), "mct": (52, 62), "mc_comp": (72, 81), "warp_emu": (81, 85), "mc_scaled": (42, 52), "itx": (85, 408), "lf": (408, 412), "cdef": (412, 416), "lr": (416, 421)}
rl = util.ref_lib()
rl.dav1d_ref_dsp_context.restype = C.c_void_p; rl.dav1d_ref_dsp_context.argtypes = [C.c_int]
hip_tab = (C.c_void_p * 421)()
assert ctx.lib.dav1d_hip_dsp_init_8bpc(hip_tab) == 0
ref_tab = (C.c_void_p * 421).from_address(rl.dav1d_ref_dsp_context(bpc))


def run(tab):
    rf = lu.RefFrame(w, h, 1, bpc, is_inter=True, sb128=False, tile_cols=1, tile_rows=1, filters=filters)
    sp = lu.default_synth(seed, n_refs=3, far_mv_pct=2)
    lu.synth(ctx, rf, sp); lu.fill_pictures(rf, seed + 1); rf.build_filter_inputs(seed)
    if tab is not None:
        rf.lib.dav1d_ref_frame_use_dsp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        assert rf.lib.dav1d_ref_frame_use_dsp(rf.h, tab, C.sizeof(tab)) == 0
    rf.recon(1)
    rec = [rf.plane(0, pl).copy() for pl in range(3)]
    rf.filter()
    fil = [rf.plane(0, pl).copy() for pl in range(3)]
    rf.destroy()
    return rec, fil


want = run(None)
COMBOS = [("all", list(RANGES))] + [("all-but-" + n, [m for m in RANGES if m != n]) for n in RANGES] + [("all again", list(RANGES))]
for name, members in COMBOS:
    tab = (C.c_void_p * 421)(*[ref_tab[k] for k in range(421)])
    for m in members:
        for k in range(*RANGES[m]):
            tab[k] = hip_tab[k]
    t0 = time.time()
    got = run(tab)
    msg = []
    for stage, idx in (("recon", 0), ("filtered", 1)):
        for pl in range(3):
            vh, vw = (h, w) if pl == 0 else (h // 2, w // 2)
            bad = np.argwhere(want[idx][pl][:vh, :vw] != got[idx][pl][:vh, :vw])
            if len(bad):
                msg.append("%s pl%d: %d px, first %s, y %d..%d x %d..%d" % (stage, pl, len(bad), bad[0].tolist(), bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max()))
    print("%-10s %.1fs %s" % (name, time.time() - t0, "; ".join(msg) if msg else "ok"), flush=True)
