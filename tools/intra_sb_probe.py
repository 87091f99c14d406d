"""A/B of the intra routes on the GPU (tools/, not part of the bench contract):
  * an 8K key frame end to end (hand-off arrays -> lister -> device) with option intra_sb = 2 (superblock by superblock) and 0 (the
    dataflow launch), each checked against the reference's own pass 2;
  * the intra pass of the bench's inter frame (1/9 of the 64x64 regions intra) launch by launch, as a graph, and superblock by superblock.
    python tools/intra_sb_probe.py [--no-check] [--size 7680x4320]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--size", default="7680x4320")
    ap.add_argument("--tiles", default="16x8")
    ap.add_argument("--modes", default="2,0")
    ap.add_argument("--lds", default="1,0", help="intra_sb_lds values to try with intra_sb = 2")
    ap.add_argument("--waves", default="0", help="intra_sb_waves values to try with intra_sb = 2 (0: the default)")
    ap.add_argument("--flow", default="0", help="intra_sb_flow values to try (1: every level in one launch)")
    ap.add_argument("--no-pass", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--sizes", action="store_true", help="the intra pass with every region of ONE block size, per size: time per step of each route")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    tc, tr = (int(v) for v in a.tiles.split("x"))
    from dav1d_amd import api
    import synth_frames as synth
    import e2e
    import lister_util as lu
    import test_postchain
    import torch
    torch.cuda.set_device(0)
    ctx = api.Context(0)
    ctx.backend = "hip"
    out = {}
    ldss = [int(v) for v in a.lds.split(",")]
    wvs = [int(v) for v in a.waves.split(",")]
    flows = [int(v) for v in a.flow.split(",")]
    combos = [(m, l, wv, fl) for m in [int(v) for v in a.modes.split(",")] for l in (ldss if m else ldss[:1]) for wv in (wvs if m and not l else wvs[:1])
              for fl in (flows if m and not l else [0])]
    for mode, lds, wv, fl in ([] if a.no_e2e else combos):
        assert ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb_flow", fl) == 0
        assert ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb", mode) == 0
        assert ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb_lds", lds) == 0
        assert ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb_waves", wv) == 0
        tag = "intra_sb_%d_lds_%d_w%d_flow%d" % (mode, lds, wv, fl) if mode else "intra_sb_0"
        chk = None if a.no_check else (lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs, is_inter=False))
        out["key_frame_" + tag] = e2e.run(ctx, w, h, 10, frames=4, threads=64, tile_cols=tc, tile_rows=tr, key_frame=True, seed=0xE2F, check=chk)
        chk = None if a.no_check else (lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs, is_inter=True))
        out["inter_10pct_" + tag] = e2e.run(ctx, w, h, 10, frames=4, threads=64, tile_cols=tc, tile_rows=tr, intra_pct=10, seed=0xE30, check=chk)
    if not a.no_pass:
        frame = synth.make_frame(w, h, 10, seed=1)
        intra = synth.make_intra_pass(frame, seed=0x1A7)
        rng = np.random.default_rng(3)
        planes = synth.make_planes(rng, w, h, 10, smooth=True)
        pics = {}
        res = {}
        for name, kw in (("enqueued", {}), ("graph", {"graph": True}), ("superblocks_lds_w8", {"sb": True}),
                         ("superblocks_l2_w4", {"sb": True}), ("superblocks_l2_w8", {"sb": True})):
            ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb_lds", 0 if "_l2" in name else 1)
            ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb_waves", int(name.rsplit("w", 1)[1]) if "_w" in name else 0)
            for rep in range(2):
                pic = ctx.picture(w, h, api.LAYOUT_I420, 10)
                for pl in range(3):
                    pic.upload(pl, planes[pl])
                res[name] = round(test_postchain.hip_intra(ctx, intra, pic, timed=True, **kw), 4)
                pics[name] = [pic.download(pl) for pl in range(3)]
                pic.free()
        res["equal"] = all(np.array_equal(pics["enqueued"][pl], pics[k][pl]) for k in pics for pl in range(3))
        res["levels"] = int(test_postchain.hip_intra.sb_levels)
        res["n_superblocks"] = int(test_postchain.hip_intra.sb_superblocks)
        res["steps"] = len(intra.batches)
        out["intra_pass_ms"] = res
    if a.sizes:
        res = {}
        for ci, sz in enumerate((64, 32, 16, 8, 4)):
            mix = [0.0] * 5
            mix[ci] = 1.0
            frame = synth.make_frame(3840, 2160, 10, seed=5, mix=tuple(mix))
            intra = synth.make_intra_pass(frame, seed=0x1A7)
            planes = synth.make_planes(np.random.default_rng(3), 3840, 2160, 10, smooth=True)
            r = {"steps": len(intra.batches), "blocks": intra.n_blocks}
            for name, kw in (("graph", {"graph": True}), ("superblocks", {"sb": True}), ("superblocks_l2", {"sb": True})):
                ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb_lds", 0 if name.endswith("l2") else 1)
                for rep in range(2):
                    pic = ctx.picture(3840, 2160, api.LAYOUT_I420, 10)
                    for pl in range(3):
                        pic.upload(pl, planes[pl])
                    r[name] = round(test_postchain.hip_intra(ctx, intra, pic, timed=True, **kw), 4)
                    pic.free()
            r["sb_us_per_step"] = round(r["superblocks"] * 1e3 / r["steps"], 2)
            res[str(sz)] = r
        out["uniform_sizes_4k"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
