"""Deblocking masks, level cache, noskip_mask and tile-edge contexts built on the device from the hand-off arrays
(dav1d_hip_lf_rects + dav1d_hip_lf_masks_build, host/lf_rects.c + csrc/lfmask.hip) against what the reference's OWN
dav1d_create_lf_mask_intra / _inter (src/lf_mask.c:259-383) leave in a real Dav1dFrameContext when they are called for the very
same blocks the way pass 1 calls them (oracle/ref_frame.c dav1d_ref_frame_build_filter_inputs)."""
import ctypes as C

import numpy as np
import pytest

import util
import lister_util as lu
from dav1d_amd import _lib

pytestmark = pytest.mark.skipif(util.ref_lib() is None, reason="needs the reference build oracle/_ref")

AV1_FILTER = np.dtype([("filter_y", "<u2", (2, 32, 3, 2)), ("filter_uv", "<u2", (2, 32, 2, 2)), ("cdef_idx", "i1", (4,)),
                       ("noskip_mask", "<u2", (16, 2))])
assert AV1_FILTER.itemsize == 1348
LF_RECT = np.dtype([("x4", "<u2"), ("y4", "<u2"), ("w4", "u1"), ("h4", "u1"), ("cls", "u1"), ("kind", "u1"), ("lvl", "u1", (2,)),
                    ("pad", "u1", (2,))])
assert LF_RECT.itemsize == 12

LF = dict(lf=(20, 28, 16, 24, 0, False))
LF_DELTAS = dict(lf=(12, 40, 30, 9, 3, True))
# segmentation: four segments with level deltas of their own, two of them lossless (4x4 WHT blocks; the mask builder's caller
# substitutes TX_4X4 for the sizes of their inter blocks, src/decode.c:1889-1893)
SEGMENTS = dict(delta_lf=[[0, 0, 0, 0], [10, -8, 6, -4], [-12, 5, 0, 9], [20, 20, -10, -10]], lossless=[0, 1, 0, 1])

CASES = [
    ("420_8", 320, 200, 1, 8, LF, {}),
    ("420_deltas_tiles", 328, 204, 1, 10, LF_DELTAS, dict(tiles=(2, 2))),
    ("sb64_tiles", 264, 200, 1, 8, LF, dict(sb128=False, tiles=(2, 3))),
    ("444_tiles", 256, 136, 3, 10, LF_DELTAS, dict(tiles=(2, 1))),
    ("422", 260, 140, 2, 8, LF, {}),
    ("400", 256, 136, 0, 8, LF, {}),
    ("key_frame", 320, 200, 1, 8, LF_DELTAS, dict(is_inter=False, tiles=(1, 2))),
    ("skips_and_splits", 384, 264, 1, 10, LF, dict(skip_pct=50, tx_split_pct=60, tiles=(3, 2))),
    # delta_lf: every superblock parsed with level deltas of its own (four of them / one for all), tables from dav1d_calc_lf_values
    ("delta_lf_multi", 520, 392, 1, 8, LF_DELTAS, dict(delta_lf=1, tiles=(2, 2))),
    ("delta_lf_single_sb64_444", 328, 264, 3, 10, LF, dict(delta_lf=2, sb128=False)),
    ("segments_lossless_skips", 392, 264, 1, 10, LF, dict(segments=SEGMENTS, n_segs=4, skip_pct=45, skip_mode_pct=15, tiles=(2, 2))),
    ("segments_lossless_444_sb64_deltas", 264, 200, 3, 8, LF_DELTAS, dict(segments=SEGMENTS, n_segs=4, skip_pct=30, sb128=False)),
    ("segments_lossless_key_422", 260, 140, 2, 10, LF, dict(segments=SEGMENTS, n_segs=3, is_inter=False)),
]


@pytest.mark.parametrize("name,w,h,layout,bpc,filters,kw", CASES, ids=[c[0] for c in CASES])
def test_device_built_masks_equal_the_reference_builders(ctx, name, w, h, layout, bpc, filters, kw):
    kw = dict(kw)
    tiles = kw.pop("tiles", (1, 1))
    rf = lu.RefFrame(w, h, layout, bpc, is_inter=kw.pop("is_inter", True), tile_cols=tiles[0], tile_rows=tiles[1], sb128=kw.pop("sb128", True),
                     filters=filters, delta_lf=kw.pop("delta_lf", 0), segments=kw.pop("segments", None))
    try:
        sp = lu.default_synth(77, **kw)
        d = lu.synth(ctx, rf, sp)
        rf.build_filter_inputs(5)
        want = rf.array("lf_mask", np.uint8).view(AV1_FILTER)
        lflvl = rf.array("lflvl", np.uint8)
        assert len(lflvl) == 8 * 4 * 8 * 2 and lflvl.any()
        # ---- ours
        rects_p, n = C.c_void_p(), C.c_size_t()
        sbt = rf.array("sb_lflvl", np.uint8)
        if rf.p.delta_lf:
            assert sbt is not None and len(sbt) and len(np.unique(sbt.reshape(-1, 512), axis=0)) > 1, "one table for every superblock: vacuous case"
        assert ctx.lib.dav1d_hip_lf_rects_sb(C.byref(d), lflvl.ctypes.data, sbt.ctypes.data if sbt is not None and len(sbt) else None,
                                             C.byref(rects_p), C.byref(n)) == 0
        assert n.value > 0
        if rf.p.seg_enabled:
            # not a vacuous case: segments carry different levels, and (inter frames) skipped inter blocks of lossless segments exist
            # whose recorded transform size is not 4x4 (Av1Block bytes: 3 intra, 4 seg_id, 6 skip, 7 uvtx; block origins have bl | bs != 0)
            assert len(np.unique(lflvl.reshape(8, -1)[:4], axis=0)) == 4
            blk = rf.array("b", np.uint8).reshape(-1, 32)
            at_origin = (blk[:, 0] | blk[:, 1]) != 0
            assert len(np.unique(blk[at_origin, 4])) == sp.n_segs
            if rf.is_inter:
                hit = at_origin & (blk[:, 3] == 0) & (blk[:, 6] == 1) & (np.asarray(rf.p.seg_lossless[:])[blk[:, 4] & 7] == 1) & (blk[:, 7] != 0)
                assert int(hit.sum()) > 10, "no skipped inter block in a lossless segment"
        ss_hor, ss_ver = int(layout != 3), int(layout == 1)
        w4, h4 = (w + 3) >> 2, (h + 3) >> 2
        bw, bh = ((w + 7) >> 3) << 1, ((h + 7) >> 3) << 1
        sb128w, sb128h = (bw + 31) >> 5, (bh + 31) >> 5
        align_h = (bh + 31) & ~31
        ntc, ntr = d.n_tile_cols, d.n_tile_rows
        got = np.zeros(sb128w * sb128h, AV1_FILTER)
        level = ctx.buffer(sb128h * 32 * d.b4_stride * 4 + 64)
        level.zero()
        r_y, r_uv = np.zeros(align_h * ntc, np.uint8), np.zeros(align_h * ntc, np.uint8)
        a_y, a_uv = np.zeros(ntr * sb128w * 32, np.uint8), np.zeros(ntr * sb128w * 32, np.uint8)
        right = (C.c_void_p * 2)(r_y.ctypes.data, r_uv.ctypes.data)
        rc = ctx.lib.dav1d_hip_lf_masks_build(ctx.h, C.byref(d), rects_p, n.value, got.ctypes.data, level.ptr, right, a_y.ctypes.data, a_uv.ctypes.data)
        ctx.lib.dav1d_hip_lf_rects_free(rects_p)
        assert rc == 0, rc
        # ---- masks
        for f in ("filter_y", "filter_uv", "noskip_mask"):
            if layout == 0 and f == "filter_uv":
                continue
            bad = np.argwhere(got[f] != want[f])
            assert not len(bad), "%s differs at (sb128, ...) %s: got %#x want %#x" % (f, bad[0], got[f][tuple(bad[0])], want[f][tuple(bad[0])])
        assert want["filter_y"].any() and want["noskip_mask"].any()
        # ---- level cache: every cell of the frame, luma entries at luma cells, chroma entries at chroma cells
        lv_want = rf.array("lf_level", np.uint8).reshape(-1, d.b4_stride, 4)
        lv_got = level.download(np.uint8, sb128h * 32 * d.b4_stride * 4).reshape(-1, d.b4_stride, 4)
        assert np.array_equal(lv_got[:h4, :w4, :2], lv_want[:h4, :w4, :2])
        if layout:
            cw4, ch4 = (w4 + ss_hor) >> ss_hor, (h4 + ss_ver) >> ss_ver
            assert np.array_equal(lv_got[:ch4, :cw4, 2:], lv_want[:ch4, :cw4, 2:])
        # ---- contexts at tile edges (what the sbrow drivers' fix-ups read): right edge of every tile column but the last,
        # bottom row of every tile row but the last
        re0, re1 = rf.array("tx_lpf_right_edge0", np.uint8), rf.array("tx_lpf_right_edge1", np.uint8)
        for tc in range(ntc - 1):
            assert np.array_equal(r_y[align_h * tc:align_h * tc + h4], re0[align_h * tc:align_h * tc + h4]), tc
            if layout:
                ah = align_h >> ss_ver
                assert np.array_equal(r_uv[ah * tc:ah * tc + ((h4 + ss_ver) >> ss_ver)], re1[ah * tc:ah * tc + ((h4 + ss_ver) >> ss_ver)]), tc
        fd = rf.filter_desc()
        a_ref = rf.array("a", np.uint8)
        base = rf.ptr("a")[0]
        for tr in range(ntr - 1):
            for x in range(w4):
                o = (tr * sb128w + (x >> 5)) * fd.a_stride
                assert a_y[(tr * sb128w + (x >> 5)) * 32 + (x & 31)] == a_ref[fd.a_tx_lpf_y - base + o + (x & 31)], (tr, x)
            if layout:
                for cx in range((w4 + ss_hor) >> ss_hor):
                    xl = cx << ss_hor
                    o = (tr * sb128w + (xl >> 5)) * fd.a_stride
                    assert a_uv[(tr * sb128w + (xl >> 5)) * 32 + ((xl & 31) >> ss_hor)] == a_ref[fd.a_tx_lpf_uv - base + o + ((xl & 31) >> ss_hor)], (tr, cx)
        level.free()
    finally:
        rf.destroy()
