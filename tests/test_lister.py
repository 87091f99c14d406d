"""The pass-2 lister against the reference's OWN pass 2.

One synthetic pass-1 output (Av1Block / cbi / cf / palettes, dav1d_synth_frame) sits in the per-frame arrays of a real
Dav1dFrameContext of the reference build (oracle/ref_frame.c).  The reference reconstructs from it on the CPU with
dav1d_decode_tile_sbrow(pass 2) -> decode_sb -> decode_b -> dav1d_recon_b_intra / dav1d_recon_b_inter; the product lists
the same arrays (dav1d_hip_lister_*), submits the tasks to a frame (dav1d_hip_frame_*) and runs the kernels.  Planes must be
byte-identical.  Covers 8 / 10 / 12 bpc, 4:0:0 / 4:2:0 / 4:2:2 / 4:4:4, 64- and 128-pixel superblocks, tiles (listed from
several threads), key frames, and every prediction tool: OBMC, compound avg / weighted / wedge / segment, local and global
warps, inter-intra, scaled references, sub8x8 chroma, palette, CfL, filter-intra, transform splitting, all transform types."""
import ctypes as C

import numpy as np
import pytest

import util
import lister_util as lu
from dav1d_amd import _lib

pytestmark = pytest.mark.skipif(util.ref_lib() is None, reason="needs the reference build oracle/_ref")

GMV = [None] * 7
GMV[0] = (3, [12345, -23456, 65536 + 800, 300, -250, 65536 - 600])
GMV[2] = (2, [-8000, 4000, 65536 + 500, 400, -400, 65536 + 500])
GMV[4] = (1, [16000, -8000, 65536, 0, 0, 65536])


def scaled_refs(w, h):
    rs = [(w, h)] * 7
    rs[1] = (2 * w, 2 * h)
    rs[3] = (w * 3 // 4 & ~7, h * 3 // 4 & ~7)
    rs[5] = (w * 3 // 2 & ~7, h + 8)
    return rs


def run_case(ctx, w, h, layout, bpc, seed, is_inter=True, tiles=(1, 1), threads=1, sb128=True, ref_sizes=None, gmv=None, segments=None,
             packed=False, interleave=False, **kw):
    rf = lu.RefFrame(w, h, layout, bpc, is_inter=is_inter, tile_cols=tiles[0], tile_rows=tiles[1], sb128=sb128,
                     screen_content=1 if kw.get("palette") else 0, ref_sizes=ref_sizes, gmv=gmv, segments=segments)
    try:
        sp = lu.default_synth(seed, **kw)
        d = lu.synth(ctx, rf, sp)
        if kw.get("intrabc_pct"):       # the case is about intra block copies: there must be some (Av1Block byte 3 = intra, byte 1 = bs)
            blk = rf.array("b", np.uint8).reshape(-1, 32)
            assert int(((blk[:, 3] == 0) & ((blk[:, 0] | blk[:, 1]) != 0)).sum()) > 50
        lu.fill_pictures(rf, seed + 1)
        rf.recon()
        got, st = lu.run_hip(ctx, rf, d, threads, packed=packed, interleave=interleave)
        bad = lu.compare(rf, got)
        assert not bad, "planes differ from the reference's pass 2: (plane, pixels, first y, x, want, got) %s" % bad
        # the coefficient arena is consumed exactly as the reference consumes it (itx zeroes what it read)
        after_ref = rf.array("cf", np.uint8)
        assert np.array_equal(st["coef_after"][:len(after_ref)], after_ref), "coefficient arena differs after the frame"
        return st
    finally:
        rf.destroy()


def test_hand_off_struct_layouts_match_the_reference():
    lib = lu.ref_lib()
    out = (C.c_int * 64)()
    lib.dav1d_ref_layouts(out)
    v = list(out)
    v = v[:v.index(-1)]
    B = _lib_av1block()
    want = [C.sizeof(B)]
    for path in ("bl", "bs", "bp", "intra", "seg_id", "skip_mode", "skip", "uvtx", "y_mode", "uv_mode", "tx", "pal_sz", "y_angle",
                 "uv_angle", "cfl_alpha", "mv", "wedge_idx", "mask_sign", "interintra_mode", "mv2d", "matrix", "comp_type", "inter_mode",
                 "motion_mode", "drl_idx", "ref", "max_ytx", "filter2d", "interintra_type", "tx_split0", "tx_split1"):
        want.append(_offset(B, path))
    want += [C.sizeof(_lib.WarpParams), _lib.WarpParams.type.offset, _lib.WarpParams.matrix.offset, _lib.WarpParams.abcd.offset]
    assert v[:len(want)] == want


class _Intra(C.Structure):
    _fields_ = [("y_mode", C.c_uint8), ("uv_mode", C.c_uint8), ("tx", C.c_uint8), ("pal_sz", C.c_uint8 * 2), ("y_angle", C.c_int8),
                ("uv_angle", C.c_int8), ("cfl_alpha", C.c_int8 * 2)]


class _M(C.Structure):
    _fields_ = [("mv", (C.c_int16 * 2) * 2), ("wedge_idx", C.c_uint8), ("mask_sign", C.c_uint8), ("interintra_mode", C.c_uint8)]


class _W(C.Structure):
    _fields_ = [("mv2d", C.c_int16 * 2), ("matrix", C.c_int16 * 4)]


class _MU(C.Union):
    _fields_ = [("m", _M), ("w", _W)]


class _Inter(C.Structure):
    _fields_ = [("u", _MU), ("comp_type", C.c_uint8), ("inter_mode", C.c_uint8), ("motion_mode", C.c_uint8), ("drl_idx", C.c_uint8),
                ("ref", C.c_int8 * 2), ("max_ytx", C.c_uint8), ("filter2d", C.c_uint8), ("interintra_type", C.c_uint8),
                ("tx_split0", C.c_uint8), ("tx_split1", C.c_uint16)]


class _BU(C.Union):
    _fields_ = [("i", _Intra), ("p", _Inter)]


def _lib_av1block():
    class B(C.Structure):          # mirrors Dav1dHipAv1Block of include/dav1d_hip.h member by member
        _fields_ = [("bl", C.c_uint8), ("bs", C.c_uint8), ("bp", C.c_uint8), ("intra", C.c_uint8), ("seg_id", C.c_uint8),
                    ("skip_mode", C.c_uint8), ("skip", C.c_uint8), ("uvtx", C.c_uint8), ("u", _BU)]
    return B


def _offset(B, name):
    if hasattr(B, name):
        return getattr(B, name).offset
    base = B.u.offset
    if hasattr(_Intra, name):
        return base + getattr(_Intra, name).offset
    if hasattr(_Inter, name):
        return base + getattr(_Inter, name).offset
    if hasattr(_M, name):
        return base + getattr(_M, name).offset
    return base + getattr(_W, name).offset


SEGMENTS = dict(delta_lf=[[0, 0, 0, 0], [10, -8, 6, -4], [-12, 5, 0, 9], [20, 20, -10, -10]], lossless=[0, 1, 0, 1])

PLAIN = dict(intra_pct=0, compound_pct=0, global_pct=0, interintra_pct=0, obmc_pct=0, warp_pct=0, tx_split_pct=0, alt_txtp_pct=0, rect_pct=0)

# (name, w, h, layout, bpc, keyword arguments)
TOOLS = [
    ("plain", 256, 192, 1, 8, dict(PLAIN)),
    ("partitions", 256, 192, 1, 8, dict(PLAIN, rect_pct=70)),
    ("vartx_txtp", 256, 192, 1, 8, dict(PLAIN, tx_split_pct=40, alt_txtp_pct=60)),
    ("compound", 256, 192, 1, 10, dict(PLAIN, compound_pct=70, masked_compound=1)),
    ("obmc", 256, 192, 1, 8, dict(PLAIN, obmc_pct=70, rect_pct=40)),
    ("local_warp", 256, 192, 1, 10, dict(PLAIN, warp_pct=70)),
    ("interintra", 256, 192, 1, 8, dict(PLAIN, interintra_pct=80, intra_pct=10)),
    ("intra_tools", 256, 192, 1, 10, dict(PLAIN, intra_pct=50, cfl_pct=50, filter_intra_pct=40, rect_pct=40)),
]


@pytest.mark.parametrize("name,w,h,layout,bpc,kw", TOOLS, ids=[t[0] for t in TOOLS])
def test_tools_one_by_one(ctx, name, w, h, layout, bpc, kw, twin_refs):
    run_case(ctx, w, h, layout, bpc, 3 + len(name), **kw)


MIX = [
    ("420_8", 256, 192, 1, 8, {}),
    ("420_10", 256, 192, 1, 10, {}),
    ("420_12_cut", 200, 136, 1, 12, {}),
    ("444_8", 256, 192, 3, 8, {}),
    ("444_10", 192, 192, 3, 10, {}),
    ("422_10", 256, 192, 2, 10, {}),
    ("400_8", 256, 192, 0, 8, {}),
    ("tiles_2x2", 320, 200, 1, 8, dict(tiles=(2, 2))),
    ("tiles_3_threads", 320, 200, 1, 10, dict(tiles=(3, 1), threads=3)),
    ("sb64_tiles", 264, 136, 1, 8, dict(sb128=False, tiles=(2, 1))),
    ("key_420_8", 256, 192, 1, 8, dict(is_inter=False)),
    ("key_444_10", 192, 128, 3, 10, dict(is_inter=False)),
    ("key_palette", 256, 192, 1, 8, dict(is_inter=False, palette=40)),
    # intra block copy (recon_b_inter on a key frame, src/recon_tmpl.c:1583-1597): sources above and to the left, odd vectors
    # (chroma half positions in 4:2:0 / 4:2:2), 4-wide / 4-high blocks carrying the chroma of their 8x8, several tiles
    ("key_intrabc_420_8", 512, 320, 1, 8, dict(is_inter=False, intrabc_pct=45, tiles=(2, 1))),
    ("key_intrabc_444_10", 384, 256, 3, 10, dict(is_inter=False, intrabc_pct=60)),
    ("key_intrabc_422_12_sb64", 520, 264, 2, 12, dict(is_inter=False, intrabc_pct=50, sb128=False, palette=20)),
    ("inter_palette_10", 256, 192, 1, 10, dict(palette=40, intra_pct=50)),
    ("global_motion", 256, 192, 1, 8, dict(gmv=GMV, global_pct=40)),
    ("global_motion_444", 192, 192, 3, 10, dict(gmv=GMV, global_pct=40)),
    ("scaled_refs", 256, 192, 1, 8, dict(ref_sizes=scaled_refs(256, 192))),
    ("scaled_refs_444_10", 192, 128, 3, 10, dict(ref_sizes=scaled_refs(192, 128))),
    # segmentation: seg_id varies from block to block, two lossless segments (their blocks: 4x4 Walsh-Hadamard transforms only,
    # src/recon_tmpl.c:347-360, src/decode.c:456-459, 1186-1188), skip_mode blocks (src/decode.c:1399-1404)
    ("segments_lossless_420_10", 256, 192, 1, 10, dict(segments=SEGMENTS, n_segs=4, skip_mode_pct=15)),
    ("segments_lossless_key_444_8", 192, 128, 3, 8, dict(segments=SEGMENTS, n_segs=4, is_inter=False)),
    ("segments_lossless_422_12_tiles", 264, 136, 2, 12, dict(segments=SEGMENTS, n_segs=3, skip_mode_pct=25, tiles=(2, 1))),
]
# the SIMT-emulated kernels are slow: the CPU run takes a cross-section, the GPU run everything
MIX_CPU = {"420_8", "420_12_cut", "444_10", "422_10", "400_8", "tiles_3_threads", "sb64_tiles", "key_444_10", "key_palette",
           "key_intrabc_420_8", "key_intrabc_444_10", "key_intrabc_422_12_sb64",
           "global_motion", "scaled_refs_444_10", "segments_lossless_420_10", "segments_lossless_key_444_8", "segments_lossless_422_12_tiles"}


@pytest.mark.parametrize("name,w,h,layout,bpc,kw", MIX, ids=[t[0] for t in MIX])
def test_every_tool_mixed(ctx, name, w, h, layout, bpc, kw):
    if ctx.backend == "emu" and name not in MIX_CPU:
        pytest.skip("GPU run only")
    ctx.set_option("chunk_order", MIX.index((name, w, h, layout, bpc, kw)) & 1)      # every other case with the lists ordered for the device
    st = run_case(ctx, w, h, layout, bpc, 100 + MIX.index((name, w, h, layout, bpc, kw)), **kw)
    if not kw.get("is_inter", True):
        assert st["steps"] > 20          # a key frame is one long wavefront


# The packing lister (Dav1dHipFrameDesc.cf): the same pixels from eob + 1 values per block, and the host arena left as the reference's
# inverse transforms leave it (zero).  The CPU run takes one case per coefficient order and pixel type; the GPU run the whole mix.
PACKED_CPU = {"420_10", "444_8", "tiles_3_threads", "key_444_10", "key_intrabc_420_8", "segments_lossless_420_10"}


@pytest.mark.parametrize("name,w,h,layout,bpc,kw", MIX + [("vartx_txtp", 256, 192, 1, 8, dict(PLAIN, tx_split_pct=40, alt_txtp_pct=60)),
                                                         ("vartx_txtp_10", 256, 192, 1, 10, dict(PLAIN, tx_split_pct=40, alt_txtp_pct=80))],
                         ids=[t[0] for t in MIX] + ["vartx_txtp", "vartx_txtp_10"])
def test_packing_lister(ctx, name, w, h, layout, bpc, kw):
    if ctx.backend == "emu" and name not in PACKED_CPU and not name.startswith("vartx"):
        pytest.skip("GPU run only")
    st = run_case(ctx, w, h, layout, bpc, 300 + len(name), packed=True, **kw)
    assert not st["coef_after"].any()


def test_cfl_of_a_palette_block_waits_for_the_luma_it_borrows(ctx):
    """A 16x4 block on an odd row carries the chroma of its 8x8: CfL averages the luma of the block above as well
    (src/recon_tmpl.c:1367-1381).  With a palette as its own luma prediction (no edges, wavefront step 1) nothing else orders it
    behind that block: the lister has to (seed 311 puts such a block into the bottom right corner of the picture)."""
    for packed in (False, True):
        run_case(ctx, 128, 64, 1, 8, 311, is_inter=False, palette=40, packed=packed)


@pytest.mark.parametrize("packed", [False, True], ids=["dense", "packed"])
def test_chunks_that_do_not_fit_the_frames_arena(ctx, packed):
    """The first frame of a size finds the chunk arena sized by guesswork (option chunk_arena_min makes the guess tiny here): the
    tile-sbrows whose prepared lists do not fit its pinned twin keep them in slabs of their own, the arena grows at frame end, the twin
    goes up first and the late chunks after it — one of them starts inside the range the twin covers."""
    c2 = util.make_context(ctx.backend)
    c2.backend = ctx.backend
    try:
        c2.set_option("chunk_arena_min", 32768)
        run_case(c2, 512, 384, 1, 10, 77, tiles=(2, 1), threads=2, packed=packed, **PLAIN)
        c2.set_option("chunk_arena_min", 4096)
        run_case(c2, 384, 256, 1, 8, 78, packed=packed)
    finally:
        c2.close()


@pytest.mark.parametrize("bpc", [8, 10])
def test_packing_lister_key_frame_through_the_dataflow_launch(ctx, bpc):
    """a key frame deep enough for the one-launch wavefront (intra_flow.hip, more than flow_min_steps steps): its units carry PACKED
    residuals; lanes of a wave that hold no block (one unit per wave) and lanes that do pass the same barriers (itx_body.h)"""
    c2 = util.make_context(ctx.backend)
    c2.backend = ctx.backend
    try:
        c2.set_option("intra_sb", 0)            # (the default route of a frame with a tiling is superblock by superblock, below)
        st = run_case(c2, 384, 256, 1, bpc, 5, is_inter=False, tiles=(2, 1), packed=True)
    finally:
        c2.close()
    assert st["steps"] >= 200 and not st["coef_after"].any()


@pytest.mark.parametrize("lds", [1, 0, 2], ids=["lds-resident", "l2-handoff", "l2-one-launch"])
@pytest.mark.parametrize("sb128", [True, False], ids=["sb128", "sb64"])
@pytest.mark.parametrize("bpc", [8, 10])
def test_key_frame_superblock_by_superblock(ctx, bpc, sb128, lds):
    """The intra wavefront as one workgroup per superblock and one launch per level of superblocks (intra_sb.hip), both forms of the
    hand-off inside a superblock, 64- and 128-pixel superblocks, several tiles (levels stop at tile edges), palette blocks (residuals
    without a prediction of their own) and CfL; the first frame finds the pinned unit arena too small (chunk_arena_min), so some
    tile-sbrows' units take the late path.  Inter frames with scattered intra blocks run the same route (levels from what the blocks
    really read of their neighbours)."""
    c2 = util.make_context(ctx.backend)
    c2.backend = ctx.backend
    try:
        c2.set_option("intra_sb", 2)
        c2.set_option("intra_sb_lds", int(lds == 1))
        c2.set_option("intra_sb_flow", int(lds == 2))         # every level in one launch, superblocks waiting for their neighbours' flags
        c2.set_option("chunk_arena_min", 4096)
        st = run_case(c2, 448, 320, 1, bpc, 21 + bpc, is_inter=False, tiles=(2, 2), threads=2, sb128=sb128, palette=15, packed=True)
        assert st["steps"] >= 50 and not st["coef_after"].any()
        run_case(c2, 448, 320, 1, bpc, 22 + bpc, is_inter=False, tiles=(1, 1), sb128=sb128)
        run_case(c2, 448, 320, 1, bpc, 23 + bpc, is_inter=True, tiles=(2, 1), threads=2, sb128=sb128, **dict(PLAIN, intra_pct=25))
        # intra block copies are predictions of the route (DAV1D_HIP_IPRED_COPY): a superblock's level lies above those of the superblocks
        # under its copies' source windows, and in one launch the copy waits for them (the LDS-resident form leaves such frames to the L2 form)
        run_case(c2, 512, 320, 1, bpc, 25 + bpc, is_inter=False, tiles=(2, 1), threads=2, sb128=sb128, intrabc_pct=50, palette=10)
        if lds == 0 and sb128:
            # inter-intra blocks are units of the route too (prediction, blend with the inter prediction, residual) ...
            run_case(c2, 448, 320, 1, bpc, 24 + bpc, is_inter=True, tiles=(2, 1), threads=2, **dict(PLAIN, intra_pct=20, interintra_pct=40))
            # ... and with intra_sb = 1 only the long wavefronts take it: the short one of this frame runs launch by launch from the
            # same submissions
            c2.set_option("intra_sb", 1)
            run_case(c2, 448, 320, 1, bpc, 23 + bpc, is_inter=True, tiles=(2, 1), threads=2, **dict(PLAIN, intra_pct=25))
    finally:
        c2.close()


def test_tile_sbrows_interleaved_across_tile_rows_with_an_odd_row_boundary(ctx):
    """dav1d's scheduler lists tile-sbrows of different tiles in any order.  64-pixel superblocks, two tile rows meeting at superblock
    row 3 (48 cells: not a multiple of the 32 cells the maps are padded to): the first superblock row of the LOWER tile is listed before
    the upper tile's, whose clearing of its share of the cell maps must stop at its own last row (lister.c)."""
    for is_inter, kw in ((False, dict(palette=10)), (True, dict(PLAIN, intra_pct=30))):
        run_case(ctx, 448, 320, 1, 10, 77, is_inter=is_inter, tiles=(2, 2), sb128=False, interleave=True, **kw)


def test_steps_of_intra_block_copies_come_from_cells_of_their_own_tile(ctx, monkeypatch):
    """An intra block copy waits for the step of the cells under its source rectangle, which decode_b keeps inside the tile
    (src/decode.c:1290-1336).  Tile rows one superblock high: a rectangle that ends at the tile's last row has the first cell row of the
    tile BELOW behind it — cells another thread lists, or has not cleared yet (the maps are recycled uncleared; DAV1D_HIP_LISTER_POISON
    fills them with 0xffff, so a step drawn from there is refused).  Found by the 2,048-stream sweep of round 6 as a rare -ERANGE; the
    synthetic generator keeps its rectangles a pixel inside the tile, tests/test_stream.py has the streams that failed."""
    monkeypatch.setenv("DAV1D_HIP_LISTER_POISON", "1")
    kw = dict(is_inter=False, sb128=False, intrabc_pct=60)
    for seed, tiles in ((31, (2, 4)), (33, (4, 2))):
        a = run_case(ctx, 1024, 256, 1, 8, seed, tiles=tiles, **kw)
        b = run_case(ctx, 1024, 256, 1, 8, seed, tiles=tiles, interleave=True, **kw)
        assert a["steps"] == b["steps"], "seed %d: %d steps top-down, %d with the lower tiles first" % (seed, a["steps"], b["steps"])


@pytest.mark.gpu
def test_larger_frame_many_tiles_threads():
    ctx = util.make_context("hip")
    ctx.backend = "hip"
    try:
        run_case(ctx, 1920, 1080, 1, 10, 7, tiles=(4, 2), threads=8)
        run_case(ctx, 1280, 720, 1, 8, 8, tiles=(2, 1), threads=2, is_inter=False)
        run_case(ctx, 1920, 1080, 1, 10, 9, tiles=(4, 2), threads=8, packed=True)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,layout,bpc,kw", [
    (3840, 2160, 3, 10, dict(tiles=(4, 2), threads=8)),                 # 4:4:4 at 4K
    (2560, 1440, 2, 8, dict(tiles=(2, 2), threads=4)),                  # 4:2:2
    (3840, 2160, 1, 12, dict(tiles=(4, 2), threads=8, gmv=GMV, global_pct=20)),
    (1920, 1080, 3, 8, dict(tiles=(2, 1), threads=2, is_inter=False, palette=30)),
    (1920, 1080, 1, 10, dict(tiles=(2, 2), threads=4, is_inter=False, intrabc_pct=30)),
    (3840, 2160, 1, 10, dict(tiles=(4, 2), threads=8, is_inter=False, sb128=False)),      # 2040 superblocks of 64 pixels, 22 levels in one launch
    (3840, 2160, 2, 8, dict(tiles=(2, 2), threads=4, is_inter=False, sb128=True, palette=20)),
], ids=["444_10_4k", "422_8_1440p", "420_12_4k_gmv", "key_444_8_1080p_palette", "key_420_10_1080p_intrabc", "key_420_10_4k_sb64", "key_422_8_4k_sb128_palette"])
def test_every_tool_at_frame_scale(w, h, layout, bpc, kw):
    """The mix of every tool (OBMC, warp, masks, inter-intra, palette, CfL, rectangular transforms ...) on whole pictures of the
    other chroma layouts and bit depths, hand-off arrays through the lister threads to pixels, against the reference's own pass 2."""
    ctx = util.make_context("hip")
    ctx.backend = "hip"
    try:
        run_case(ctx, w, h, layout, bpc, 40 + layout + bpc, **kw)
    finally:
        ctx.close()


def test_reference_pass2_on_worker_threads_equals_one_thread():
    """oracle/ref_frame.c dav1d_ref_frame_recon_mt (the CPU peer bench.py times): tiles on a pool of workers give the picture
    the single-threaded walk gives."""
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    import util
    ctx = util.make_context("emu")
    import e2e
    outs = []
    for threads in (1, 6):
        rf = lu.RefFrame(640, 384, 1, 10, is_inter=True, sb128=True, tile_cols=4, tile_rows=3)
        sp = e2e.c2_params(5)
        sp.intra_pct = 20
        lu.synth(ctx, rf, sp)
        lu.fill_pictures(rf, 9)
        rf.recon(threads)
        outs.append([rf.plane(0, pl).copy() for pl in range(3)])
        rf.destroy()
    for pl in range(3):
        assert np.array_equal(outs[0][pl], outs[1][pl])
    ctx.close()
