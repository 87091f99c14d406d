"""The recon step of the bench's 8K 10-bit workload with its pictures in the tiled twin only (dav1d_hip_recon_list_run_tiled) under a list
of context-option sets, next to the raster layout, in ONE process on ONE box (boxes of the pool differ by 10 %): ms per step, per-kernel
event times, parity of every variant against the raster run.
    python tools/layout_sweep.py [--sets "recon_fuse=15" "recon_fuse=14,recon_pair_streams=2" ...] [--kernels]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

NAMES = ["recon_%dx%d" % (4 << k, 4 << k) for k in range(5)] + ["mc_%dx%d" % (4 << (b // 3), 4 << (b % 3)) for b in range(15)] + ["comp"] + ["itx_%d" % b for b in range(19)]
DEFAULTS = {"recon_fuse": 15, "recon_pair_streams": 2, "recon_lanes": 1, "recon_pipeline": 16384, "recon_coop_below": 4096}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--bpc", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--sets", nargs="*", default=[""])
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--no-raster", action="store_true")
    ap.add_argument("--edge-frac", type=float, default=0.05, help="share of the blocks whose motion vectors point across the picture's edge (bench.py: 0.05)")
    ap.add_argument("--phases", action="store_true", help="with a -DDV_PHASES variant (--lib): shader-clock cycles per wave and phase of every kernel, from one event-bracketed run")
    ap.add_argument("--inflight", type=int, default=1, help="frames in flight: N contexts with streams and lists of their own, the steps dealt over them in turn (tiled sets only)")
    ap.add_argument("--digest", action="store_true", help="a digest of the last picture of every set (to compare runs of different libraries)")
    ap.add_argument("--lib", default=None, help="a variant build (tools/build_variant.py) instead of dav1d_amd/libdav1d_hip.so")
    a = ap.parse_args()
    import torch
    from dav1d_amd import api
    import synth_frames as synth
    stream = torch.cuda.current_stream()
    ctx = api.Context(0, stream=stream.cuda_stream, lib_path=a.lib)
    w, h, bpc = a.width, a.height, a.bpc
    frame = synth.make_frame(w, h, bpc, seed=0xDA71D002, mv_range_px=64, edge_frac=a.edge_frac, n_refs=3)
    rng = np.random.default_rng(1234)
    ref_host = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst_host = synth.make_planes(rng, w, h, bpc, smooth=False)
    refs = []
    for rp in ref_host:
        r = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            r.upload(pl, rp[pl])
        r.retile()
        refs.append(r)
    dsts = []
    for _ in range(4):
        d = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            d.upload(pl, dst_host[pl])
        dsts.append(d)
    prep = torch.zeros(frame.prep_elems, dtype=torch.int16, device="cuda")
    tdt = torch.int16 if bpc == 8 else torch.int32
    pristine = torch.from_numpy(frame.coef).to("cuda")
    n_arena = a.steps + 10
    arenas = torch.empty((n_arena, pristine.numel()), dtype=tdt, device="cuda")

    def fresh():
        for i in range(n_arena):
            arenas[i].copy_(pristine)
        torch.cuda.synchronize()

    extra = []          # --inflight: further contexts, each on a stream of its own
    for _ in range(a.inflight - 1):
        st2 = torch.cuda.Stream()
        extra.append((api.Context(0, stream=st2.cuda_stream, lib_path=a.lib), st2, torch.zeros(frame.prep_elems, dtype=torch.int16, device="cuda")))

    def timed(rl, tiled, more=()):
        fresh()
        lanes = [(rl, prep)] + [(r2, p2) for r2, p2 in more]
        def run(i):
            l_, p_ = lanes[i % len(lanes)]
            (l_.run_tiled if tiled else l_.run)(dsts[i % 4], refs, p_.data_ptr(), arenas[i].data_ptr())
        def sync_all():
            ctx.sync()
            for c2, _, _ in extra:
                c2.sync()
            torch.cuda.synchronize()
        for i in range(4):
            run(i)
        sync_all()
        t0 = time.perf_counter()
        for i in range(4, 4 + a.steps):
            run(i)
        sync_all()
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        pics = [dsts[(3 + a.steps) % 4].download(pl) for pl in range(3)]
        return dt, pics

    def kernels(rl, tiled):
        fresh()
        ms = (C.c_float * 40)()
        cnt = (C.c_size_t * 40)()
        best = [1e9] * 40
        for rep in range(3):
            rarr = (api.Picture * len(refs))(*[r.pic for r in refs])
            fn = ctx.lib.dav1d_hip_recon_list_run_tiled_timed if tiled else ctx.lib.dav1d_hip_recon_list_run_timed
            rc = fn(ctx.h, rl.h, C.byref(dsts[0].pic), rarr, len(refs), prep.data_ptr(), None, arenas[rep].data_ptr(), ms, cnt)
            assert rc == 0, rc
            best = [min(x, y) for x, y in zip(best, ms)]
        return {NAMES[k]: round(best[k] * 1e3, 1) for k in range(40) if cnt[k]}

    def phases(rl, tiled):
        """cycles per wave by phase (common.h DV_PHASE): the bodies' slots and the kernels' own"""
        import ctypes
        raw = C.CDLL(a.lib)
        bufs = {}
        for unit in ("mc", "recon", "itx"):
            getattr(raw, "dav1d_hip_debug_phases_" + unit)(None, 1)
        fresh()
        ms = (C.c_float * 40)()
        cnt = (C.c_size_t * 40)()
        rarr = (api.Picture * len(refs))(*[r.pic for r in refs])
        fn = ctx.lib.dav1d_hip_recon_list_run_tiled_timed if tiled else ctx.lib.dav1d_hip_recon_list_run_timed
        assert fn(ctx.h, rl.h, C.byref(dsts[0].pic), rarr, len(refs), prep.data_ptr(), None, arenas[0].data_ptr(), ms, cnt) == 0
        ctx.sync()
        out = {}
        for unit in ("mc", "recon", "itx"):
            buf = (ctypes.c_ulonglong * 1024)()
            assert getattr(raw, "dav1d_hip_debug_phases_" + unit)(buf, 0) == 0
            v = list(buf)
            shapes = [(tw, th) for tw in (4, 8, 16, 32, 64) for th in (4, 8, 16)]
            for to_lds in (0, 1):
                for b, (tw, th) in enumerate(shapes):
                    base_ = b * 16 + 256 * to_lds
                    n = v[base_ + 9]
                    if n:
                        nm = ("pred_in_pair" if to_lds else "mc") + "_%dx%d" % (tw, th)
                        out["%s.%s" % (unit, nm)] = dict(zip(("records", "gather", "horizontal", "vertical", "gather2", "horizontal2", "vertical2", "combine_store", "body"),
                                                             [round(v[base_ + k] / n) for k in range(9)]), bodies=n)
            for tx in range(19):
                for pl in (0, 8):
                    base_ = 512 + tx * 16 + pl
                    n = v[base_ + 5]
                    if n:
                        d = dict(zip(("loads_landed", "rows_in_regs", "row_pass", "column_pass_store", "body"), [round(v[base_ + k] / n) for k in range(5)]), bodies=n)
                        if not pl and v[base_ + 7]:
                            d["tile_write_out"] = round(v[base_ + 7] / n)
                        out["%s.itx%s_%d" % (unit, "_in_pair" if pl else "", tx)] = d
            for cls in range(5):
                base_ = 768 + cls * 16
                n = v[base_ + 4]
                if n:
                    out["%s.pair_%dx%d" % (unit, 4 << cls, 4 << cls)] = dict(zip(("predictions", "transform", "tile_write_out", "wave"), [round(v[base_ + k] / n) for k in range(4)]), waves=n)
        return out

    base = None
    if not a.no_raster:
        ctx.set_option("ref_twin", 0)
        rl = ctx.recon_list(dsts[0], frame.mc, frame.comp, frame.itx)
        for d in dsts:
            d.pic.twin_ok = 0
        dt, base = timed(rl, False)
        o = {"layout": "raster", "ms_per_step": round(dt, 4)}
        if a.kernels:
            o["kernels_us"] = kernels(rl, False)
        if a.phases:
            o["cycles_per_wave"] = phases(rl, False)
        print(json.dumps(o), flush=True)
        rl.destroy()
        ctx.set_option("ref_twin", 1)
    for st in a.sets:
        opts = dict(DEFAULTS)
        for kv in [x for x in st.split(",") if x]:
            k, v = kv.split("=")
            opts[k] = int(v)
        for k, v in opts.items():
            ctx.set_option(k, v)
        rl = ctx.recon_list(dsts[0], frame.mc, frame.comp, frame.itx)        # (the pairing is decided at list creation)
        more = []
        for c2, _, p2 in extra:
            for k, v in opts.items():
                c2.set_option(k, v)
            more.append((c2.recon_list(dsts[0], frame.mc, frame.comp, frame.itx), p2))
        for d in dsts:
            d.pic.twin_ok = 0
            for pl in range(3):
                d.upload(pl, dst_host[pl])
        dt, pics = timed(rl, True, more)
        for r2, _ in more:
            r2.destroy()
        o = {"layout": "tiled", "lib": os.path.basename(a.lib) if a.lib else None, "opts": st, "inflight": a.inflight, "ms_per_step": round(dt, 4), "twin_only": int(dsts[0].pic.twin_ok)}
        if base is not None:
            o["equals_first" if a.no_raster else "equals_raster"] = all(np.array_equal(base[pl], pics[pl]) for pl in range(3))
        elif a.no_raster:
            base = pics         # (without the raster run every set is compared with the first one)
        if a.digest:
            import hashlib
            o["digest"] = hashlib.sha1(b"".join(np.ascontiguousarray(p_).tobytes() for p_ in pics)).hexdigest()[:12]
        if a.kernels:
            o["kernels_us"] = kernels(rl, True)
        if a.phases:
            o["cycles_per_wave"] = phases(rl, True)
        print(json.dumps(o), flush=True)
        rl.destroy()


if __name__ == "__main__":
    main()
