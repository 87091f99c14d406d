"""The filter lister (deblocking masks / CDEF indices / restoration units -> task records) against the reference's OWN
in-loop filters: dav1d_filter_sbrow_{8,16}bpc (src/recon_tmpl.c:2100-2109: dav1d_loopfilter_sbrow_cols / _rows with the
tile-edge mask fix-ups, dav1d_copy_lpf, dav1d_cdef_brow, dav1d_lr_sbrow) run superblock row by superblock row on a real
Dav1dFrameContext whose masks and level cache were built by the reference's dav1d_create_lf_mask_intra / _inter for the very
blocks of the frame.  The device side: dav1d_hip_lister_filter_sbrow -> dav1d_hip_frame_* -> kernels, after the lister-driven
reconstruction of the same frame.  The final pictures must be byte-identical."""
import numpy as np
import pytest

import util
import lister_util as lu

pytestmark = pytest.mark.skipif(util.ref_lib() is None, reason="needs the reference build oracle/_ref")

LF = dict(lf=(20, 28, 16, 24, 0, False))
LF_DELTAS = dict(lf=(12, 40, 30, 0, 3, True))
CDEF = dict(cdef=(5, 2, [17, 33, 0, 63], [5, 0, 20, 48]))
CDEF3 = dict(cdef=(3, 3, [4, 9, 62, 3, 0, 21, 40, 1], [0, 7, 12, 63, 16, 0, 2, 31]))
LR_SW = dict(lr=([1, 1, 1], [6, 6]))
LR_W_S = dict(lr=([2, 3, 0], [7, 6]))
LR_BIG = dict(lr=([3, 1, 2], [8, 7]))
ALL = dict(LF, **CDEF, **LR_SW)
SEGMENTS = dict(delta_lf=[[0, 0, 0, 0], [10, -8, 6, -4], [-12, 5, 0, 9], [20, 20, -10, -10]], lossless=[0, 1, 0, 1])


def run_case(ctx, w, h, layout, bpc, seed, filters, is_inter=True, tiles=(1, 1), sb128=True, own_masks=False, sr_w=0, delta_lf=0, segments=None,
             **kw):
    rf = lu.RefFrame(w, h, layout, bpc, is_inter=is_inter, tile_cols=tiles[0], tile_rows=tiles[1], sb128=sb128, filters=filters, sr_w=sr_w,
                     delta_lf=delta_lf, segments=segments)
    try:
        sp = lu.default_synth(seed, **kw)
        d = lu.synth(ctx, rf, sp)
        lu.fill_pictures(rf, seed + 1)
        rf.build_filter_inputs(seed)
        rf.recon()
        n_pl = 1 if layout == 0 else 3
        before = [rf.plane(0, pl).copy() for pl in range(n_pl)]
        rf.filter()
        if not sr_w:
            assert any(not np.array_equal(before[pl], rf.plane(0, pl)) for pl in range(n_pl)), "the filters changed nothing: vacuous case"
        else:
            assert rf.plane(8, 0)[:h, :sr_w].any(), "the upscaled picture is empty: vacuous case"
        got, _ = lu.run_hip(ctx, rf, d, 1, with_filters=True, own_masks=own_masks)
        bad = lu.compare(rf, got)
        assert not bad, "filtered planes differ from dav1d_filter_sbrow: (plane, pixels, first y, x, want, got) %s" % bad
    finally:
        rf.destroy()


CASES = [
    ("deblock", 320, 200, 1, 8, LF, {}),
    ("deblock_deltas_tiles", 320, 200, 1, 10, LF_DELTAS, dict(tiles=(2, 2))),
    ("deblock_sb64_tiles", 264, 200, 1, 8, LF, dict(sb128=False, tiles=(2, 3))),
    ("deblock_444", 256, 136, 3, 10, LF, dict(tiles=(2, 1))),
    ("deblock_422", 256, 136, 2, 8, LF, {}),
    ("deblock_400", 256, 136, 0, 8, LF, {}),
    ("cdef", 320, 200, 1, 8, CDEF, {}),
    ("cdef_8_strengths_12bit", 320, 200, 1, 12, CDEF3, {}),
    ("cdef_444", 256, 136, 3, 10, CDEF, {}),
    ("cdef_422", 256, 136, 2, 10, CDEF3, {}),
    ("cdef_skips", 320, 200, 1, 8, CDEF, dict(skip_pct=85)),
    ("lr_switchable", 320, 200, 1, 8, LR_SW, {}),
    ("lr_wiener_sgr_128", 320, 264, 1, 10, LR_W_S, {}),
    ("lr_256_units_444", 400, 264, 3, 8, LR_BIG, {}),
    ("all_tiles", 320, 200, 1, 10, ALL, dict(tiles=(2, 2))),
    ("all_key_frame", 320, 200, 1, 8, ALL, dict(is_inter=False)),
    ("all_sb64_cut", 296, 168, 1, 10, ALL, dict(sb128=False, tiles=(2, 2))),
    # super-resolution (dav1d_filter_sbrow_resize between CDEF and restoration; restoration units, their lr_mask and the stripe
    # border rows in the upscaled frame): denominators 9 .. 16 of the coded width
    ("superres_all_420_8", 320, 200, 1, 8, ALL, dict(sr_w=480, tiles=(2, 1))),
    ("superres_cdef_only_444_10", 256, 136, 3, 10, CDEF, dict(sr_w=288)),
    ("superres_lr_big_units_10", 320, 264, 1, 10, dict(LF, **LR_BIG), dict(sr_w=640)),
    ("superres_key_422_12", 264, 136, 2, 12, ALL, dict(sr_w=400, is_inter=False)),
    ("superres_width_not_8n", 324, 200, 1, 8, ALL, dict(sr_w=486)),     # the resampler reads the columns up to the 8x8 block grid
    ("delta_lf_pass1_masks", 520, 264, 1, 8, ALL, dict(delta_lf=1)),     # levels that differ from superblock to superblock
    ("segments_lossless_pass1_masks", 328, 200, 1, 10, ALL, dict(segments=SEGMENTS, n_segs=4, skip_pct=40, skip_mode_pct=15)),
]
CPU = {"deblock_deltas_tiles", "deblock_sb64_tiles", "deblock_444", "deblock_400", "cdef_8_strengths_12bit", "cdef_422", "cdef_skips",
       "lr_wiener_sgr_128", "lr_256_units_444", "all_tiles", "all_key_frame", "all_sb64_cut", "superres_all_420_8",
       "superres_cdef_only_444_10", "superres_lr_big_units_10", "superres_key_422_12", "superres_width_not_8n", "delta_lf_pass1_masks",
       "segments_lossless_pass1_masks"}


@pytest.mark.parametrize("name,w,h,layout,bpc,filters,kw", CASES, ids=[c[0] for c in CASES])
def test_filters_match_dav1d_filter_sbrow(ctx, name, w, h, layout, bpc, filters, kw):
    if ctx.backend == "emu" and name not in CPU:
        pytest.skip("GPU run only")
    run_case(ctx, w, h, layout, bpc, 40 + [c[0] for c in CASES].index(name), filters, **kw)


@pytest.mark.gpu
def test_filters_1080p():
    ctx = util.make_context("hip")
    ctx.backend = "hip"
    try:
        run_case(ctx, 1920, 1080, 1, 10, 77, ALL, tiles=(4, 2))
    finally:
        ctx.close()


OWN = [("all_tiles", 320, 200, 1, 10, ALL, dict(tiles=(2, 2))), ("deltas_444_tiles", 256, 136, 3, 10, dict(LF_DELTAS, **CDEF), dict(tiles=(2, 2))),
       ("key_frame_sb64", 296, 168, 1, 8, ALL, dict(is_inter=False, sb128=False, tiles=(2, 2))),
       ("delta_lf_tiles", 520, 264, 1, 10, dict(LF_DELTAS, **CDEF), dict(delta_lf=1, tiles=(2, 1))),
       # segments with levels of their own, two of them lossless, skipped and skip_mode blocks among them: the whole chain from masks the
       # product built with Dav1dHipFrameDesc.lossless
       ("segments_lossless_tiles", 392, 264, 1, 10, ALL, dict(segments=SEGMENTS, n_segs=4, skip_pct=40, skip_mode_pct=15, tiles=(2, 2))),
       ("segments_lossless_444_8", 256, 136, 3, 8, dict(LF_DELTAS, **CDEF), dict(segments=SEGMENTS, n_segs=4, skip_pct=35))]


@pytest.mark.parametrize("name,w,h,layout,bpc,filters,kw", OWN, ids=[c[0] for c in OWN])
def test_filters_from_masks_the_product_built_itself(ctx, name, w, h, layout, bpc, filters, kw):
    """The same chain with nothing of the reference's pass 1 but cdef_idx (which the bitstream carries): deblocking masks,
    noskip_mask, level cache and the tile-edge contexts come from dav1d_hip_lf_rects + dav1d_hip_lf_masks_build."""
    run_case(ctx, w, h, layout, bpc, 90 + [c[0] for c in OWN].index(name), filters, own_masks=True, **kw)
