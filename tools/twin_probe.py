"""A/B of the reference layout on the bench's own workload (one call, one box): the recon list of an 8K 10-bit inter frame with the
references read in raster order and through their tiled twins; per-kernel times of both (dav1d_hip_recon_list_run_timed), the
retile pass, and the pictures of both runs compared.  python tools/twin_probe.py [--width W --height H --bpc B --steps N]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--bpc", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--fuse", type=int, default=-1)
    ap.add_argument("--opt", action="append", default=[], help="context option name=value (dav1d_hip_set_option)")
    ap.add_argument("--only", default="", help="comma list of modes to run (raster, tiled, tiled+retile_dst)")
    ap.add_argument("--no-kernels", action="store_true")
    a = ap.parse_args()
    import torch
    from dav1d_amd import api
    import synth_frames as synth
    stream = torch.cuda.current_stream()
    ctx = api.Context(0, stream=stream.cuda_stream)
    if a.fuse >= 0:
        ctx.set_option("recon_fuse", a.fuse)
    for o in a.opt:
        k, v = o.split("=")
        ctx.set_option(k, int(v))
    w, h, bpc = a.width, a.height, a.bpc
    frame = synth.make_frame(w, h, bpc, seed=0xDA71D002, mv_range_px=64, edge_frac=0.05, n_refs=3)
    rng = np.random.default_rng(1234)
    ref_host = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst_host = synth.make_planes(rng, w, h, bpc, smooth=False)
    refs = []
    for rp in ref_host:
        r = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            r.upload(pl, rp[pl])
        refs.append(r)
    dsts = []
    for _ in range(4):
        d = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            d.upload(pl, dst_host[pl])
        dsts.append(d)
    rl = ctx.recon_list(dsts[0], frame.mc, frame.comp, frame.itx)
    prep = torch.zeros(frame.prep_elems, dtype=torch.int16, device="cuda")
    tdt = torch.int16 if bpc == 8 else torch.int32
    pristine = torch.from_numpy(frame.coef).to("cuda")
    n_arena = a.steps + 8
    arenas = torch.empty((n_arena, pristine.numel()), dtype=tdt, device="cuda")

    def fresh():
        for i in range(n_arena):
            arenas[i].copy_(pristine)
        torch.cuda.synchronize()

    out = {}
    pics = {}
    only = [m for m in a.only.split(",") if m]
    retiled = False
    for mode in ("raster", "tiled", "tiled+retile_dst", "tiled+retile_dst_overlapped", "tiled+twin_written_by_the_launches"):
        if only and mode not in only:
            continue
        if mode != "raster" and not retiled:
            for r in refs:
                r.retile()
            torch.cuda.synchronize()
            retiled = True
        fresh()
        direct = mode.endswith("launches")
        for i in range(3):
            (rl.run_twin if direct else rl.run)(dsts[i % 4], refs, prep.data_ptr(), arenas[i].data_ptr())
            if mode.endswith("dst"):
                dsts[i % 4].retile()
            elif mode.endswith("overlapped"):
                dsts[i % 4].retile(True)
        ctx.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + a.steps):
            (rl.run_twin if direct else rl.run)(dsts[i % 4], refs, prep.data_ptr(), arenas[i].data_ptr())
            if mode.endswith("dst"):
                dsts[i % 4].retile()
            elif mode.endswith("overlapped"):
                dsts[i % 4].retile(True)
        ctx.sync()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        out[mode] = {"ms_per_step": round(dt, 4)}
        pics[mode] = [dsts[(2 + a.steps) % 4].download(pl) for pl in range(3)]
        if direct:
            # the twin the launches wrote, against a retile of the same picture
            d = dsts[(2 + a.steps) % 4]
            import ctypes as CC
            nb = d.pic.p[0].stride * ((h + 7) & ~7)
            t1 = np.zeros(nb, np.uint8); t2 = np.zeros(nb, np.uint8)
            ctx.lib.dav1d_hip_sync(ctx.h)
            ctx.lib.dav1d_hip_download(ctx.h, t1.ctypes.data, d.pic.twin[0], nb)
            d.retile(); ctx.lib.dav1d_hip_sync(ctx.h)
            ctx.lib.dav1d_hip_download(ctx.h, t2.ctypes.data, d.pic.twin[0], nb)
            out[mode]["twin_equals_retile"] = bool(np.array_equal(t1, t2))
        if a.no_kernels or direct:
            continue
        # per-kernel
        fresh()
        ms = (C.c_float * 40)()
        cnt = (C.c_size_t * 40)()
        rarr = (api.Picture * len(refs))(*[r.pic for r in refs])
        best = [1e9] * 40
        for rep in range(3):
            rc = ctx.lib.dav1d_hip_recon_list_run_timed(ctx.h, rl.h, C.byref(dsts[0].pic), rarr, len(refs), prep.data_ptr(), None,
                                                        arenas[rep].data_ptr(), ms, cnt)
            assert rc == 0, rc
            best = [min(x, y) for x, y in zip(best, ms)]
        names = ["recon_%dx%d" % (4 << k, 4 << k) for k in range(5)] + ["mc_%dx%d" % (4 << (b // 3), 4 << (b % 3)) for b in range(15)] + ["comp"] + ["itx_%d" % b for b in range(19)]
        out[mode]["kernels_us"] = {names[k]: round(best[k] * 1e3, 1) for k in range(40) if cnt[k]}
    # the retile pass alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        dsts[0].retile()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    out["retile_ms"] = round(best, 4)
    if "raster" in pics and "tiled" in pics:
        out["parity_tiled_vs_raster"] = all(np.array_equal(pics["raster"][pl], pics["tiled"][pl]) for pl in range(3))
    if "raster" in pics and "tiled+twin_written_by_the_launches" in pics:
        out["parity_direct_vs_raster"] = all(np.array_equal(pics["raster"][pl], pics["tiled+twin_written_by_the_launches"][pl]) for pl in range(3))
    out["opts"] = a.opt + (["fuse=%d" % a.fuse] if a.fuse >= 0 else [])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
