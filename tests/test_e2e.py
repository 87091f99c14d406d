"""The end-to-end route of bench.py (tests/e2e.py): pass-1 hand-off arrays -> pass-2 lister on host threads -> chunks ->
device, for an inter frame and for a key frame (every block through the intra wavefront), checked against the reference's
own pass 2 on a real Dav1dFrameContext."""
import pytest

import lister_util as lu
import e2e


@pytest.mark.parametrize("packed", [False, True], ids=["dense-upload", "packing-lister"])
@pytest.mark.parametrize("key_frame", [False, True], ids=["inter", "key"])
def test_end_to_end_route_matches_reference_pass2(ctx, key_frame, packed, twin_refs):
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (384, 256) if ctx.backend == "emu" else (1920, 1080)
    out = e2e.run(ctx, w, h, 10, frames=2, threads=3, tile_cols=2, tile_rows=2, seed=77, key_frame=key_frame, packed=packed,
                  check=lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs, is_inter=not key_frame))
    assert out["parity"].startswith("bit-exact"), out["parity"]
    assert out["wavefront_steps"] >= (2 if key_frame else 0)


def test_full_route_with_filters_matches_the_reference(ctx):
    """lister_util.full_route_rate (bench.py's end_to_end_full_table leg): hand-off arrays + pass 1's filter inputs -> library threads
    (dav1d_hip_lister_run, dav1d_hip_lister_filter_run) -> one dav1d_hip_frame_end with every stage; the helper raises when the
    filtered picture differs from dav1d_decode_tile_sbrow + dav1d_filter_sbrow."""
    import lister_util as lu
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (384, 264) if ctx.backend == "emu" else (1920, 1080)
    out = lu.full_route_rate(ctx, w, h, 10, tile_cols=2, tile_rows=2, threads=4, frames=3)
    assert out and out["parity"].startswith("bit-exact") and out["total_ms"] > 0


def test_frames_in_flight_with_the_packing_lister(ctx):
    """e2e.run_sustained / lister_util.full_route_sustained (bench.py's *_sustained legs): frame n + 1 is listed while frame n runs on the
    device; the packing lister (Dav1dHipFrameDesc.cf) moves the coefficients that exist into the frame's own arena and leaves the host
    arena zero, nothing dense crosses the host link; the last frame of the run equals the reference's."""
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (384, 256) if ctx.backend == "emu" else (1920, 1080)
    out = e2e.run_sustained(ctx, w, h, 10, frames=4, threads=3, tile_cols=2, tile_rows=2, seed=78, warm=1,
                            check=lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs))
    assert out["parity"].startswith("bit-exact"), out["parity"]
    assert out["host_arena_left_zero"] and 0 < out["packed_coef_bytes_per_frame"] < w * h * 6 // 2
    out = lu.full_route_sustained(ctx, w, h + 8, 10, tile_cols=2, tile_rows=2, threads=4, frames=4, warm=1)
    assert out and out["parity"].startswith("bit-exact") and out["host_arena_left_zero"]


def test_packed_and_dense_residuals_do_not_mix_in_a_frame(ctx):
    """a frame's residual tasks all point into the frame's own coefficient arena (dav1d_hip_frame_submit_coefs) or all into the
    caller's dense one: dav1d_hip_frame_end refuses the mix and a packed frame that is handed a dense arena."""
    import ctypes as C
    import numpy as np
    from dav1d_amd import api
    cur = ctx.picture(64, 64, api.LAYOUT_I420, 8)
    frame = ctx.frame(cur, [])
    vals = np.array([64, 0, 0, 3], np.int16)
    base = C.c_uint32(123)
    assert ctx.lib.dav1d_hip_frame_submit_coefs(frame.h, vals.ctypes.data, len(vals), C.byref(base)) == 0 and base.value == 0
    assert ctx.lib.dav1d_hip_frame_submit_coefs(frame.h, vals.ctypes.data, len(vals), C.byref(base)) == 0 and base.value == 32   # 64-byte segments
    assert ctx.lib.dav1d_hip_frame_coef_bytes(frame.h) == 128
    t = np.zeros(2, api.ITX_TASK)
    t["tx"], t["eob"], t["flags"], t["cf_off"] = 0, 3, [1, 0], [0, 0]
    t["dst_off"] = [0, 8]
    assert ctx.lib.dav1d_hip_frame_submit_tile_sbrow(frame.h, None, 0, None, 0, t.ctypes.data, 2) == 0
    dense = ctx.buffer(4096)
    dense.zero()
    assert ctx.lib.dav1d_hip_frame_end(frame.h, None, None, None, None, None) == -22
    frame.destroy()
    frame = ctx.frame(cur, [])
    assert ctx.lib.dav1d_hip_frame_submit_coefs(frame.h, vals.ctypes.data, len(vals), C.byref(base)) == 0
    assert ctx.lib.dav1d_hip_frame_submit_tile_sbrow(frame.h, None, 0, None, 0, t[:1].ctypes.data, 1) == 0
    assert ctx.lib.dav1d_hip_frame_end(frame.h, dense.ptr, None, None, None, None) == -22
    frame.destroy()
    dense.free()
    cur.free()


@pytest.mark.parametrize("intrabc_pct", [0, 40], ids=["no-copies", "intra-block-copies"])
@pytest.mark.parametrize("waves", [0, 1, 4, 8], ids=["default", "1-wave", "4-waves", "8-waves"])
def test_key_frame_with_either_workgroup_size_of_the_superblock_launch(ctx, waves, intrabc_pct):
    """context option intra_sb_waves: the one-launch superblock form (intra_sb.hip) runs with workgroups of four waves (two superblocks per CU
    in flight: what a frame without intra block copies gets by default), of eight (the default with copies), or with ONE wave per superblock
    (its units one after the other, no barrier: the default where superblocks hold a handful of units); all must give the reference's pixels
    on both kinds of key frame"""
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (384, 256) if ctx.backend == "emu" else (1920, 1080)
    ctx.set_option("intra_sb_waves", waves)
    try:
        out = e2e.run(ctx, w, h, 10, frames=2, threads=3, tile_cols=2, tile_rows=2, seed=79, key_frame=True, intrabc_pct=intrabc_pct,
                      check=lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs, is_inter=False))
    finally:
        ctx.set_option("intra_sb_waves", 0)
    assert out["parity"].startswith("bit-exact"), out["parity"]
    assert out["wavefront_steps"] >= 2


@pytest.mark.parametrize("fail_at", [0, 1, 4], ids=["first-superblock", "second", "fifth"])
@pytest.mark.parametrize("waves", [4, 8], ids=["4-waves", "8-waves"])
def test_key_frame_whose_one_launch_pass_gives_up_is_finished_by_the_launches_per_level(ctx, waves, fail_at):
    """frame.hip: when workgroups of the one-launch superblock form give up waiting for a neighbour (context option intra_sb_fail_at makes the
    workgroup of that index do what one does whose wait ran out; everything that reads its superblock follows, some of them mid-way), the frame
    is not lost: the units the launch did reconstruct are marked, the launches per level run the others, and the pictures are the reference's.
    The context counts such frames (dav1d_hip_get_option "intra_sb_fallbacks")."""
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (384, 256) if ctx.backend == "emu" else (1920, 1080)
    before = ctx.get_option("intra_sb_fallbacks")
    ctx.set_option("intra_sb_waves", waves)
    ctx.set_option("intra_sb_fail_at", fail_at)
    try:
        out = e2e.run(ctx, w, h, 10, frames=2, threads=3, tile_cols=2, tile_rows=2, seed=81, key_frame=True,
                      check=lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs, is_inter=False))
    finally:
        ctx.set_option("intra_sb_fail_at", -1)
        ctx.set_option("intra_sb_waves", 0)
    assert out["parity"].startswith("bit-exact"), out["parity"]
    assert ctx.get_option("intra_sb_fallbacks") >= before + 2
