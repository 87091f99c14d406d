"""Whole-frame parity of the itx+mc reconstruction path: the synthetic pass-2 task lists of
tests/synth_frames.py run (a) through the HIP backend via the C ABI and (b) through the oracle's
DSP function pointers via oracle/replay.c; every output plane, the prep arena and the
coefficient arena must be byte-identical."""
import ctypes as C

import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth


class RP(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("stride", C.c_ssize_t * 3), ("w", C.c_int * 3), ("h", C.c_int * 3)]


def planes_struct(planes, w, h):
    rp = RP()
    for i, p in enumerate(planes):
        rp.data[i] = p.ctypes.data
        rp.stride[i] = p.strides[0]
        rp.w[i] = w if i == 0 else (w + 1) >> 1
        rp.h[i] = h if i == 0 else (h + 1) >> 1
    return rp


def oracle_frame(oracle, frame, dst_planes, ref_planes_list, threads=1, timing=None):
    """Replays the frame on the host; returns (planes, prep, coef) after reconstruction.  threads > 1: the phases of the
    replay spread over host threads (oracle/replay.c dav1d_replay_recon_mt; avg / w_avg compounds only).  timing: dict that receives
    the seconds spent inside the replay calls proper (without this harness's copies of the inputs)."""
    import time
    rl = util.replay_lib()
    entry = C.cast(oracle._entry, C.c_void_p)
    w, h, bpc = frame.w, frame.h, frame.bpc
    dst = synth.copy_planes(dst_planes)
    drp = planes_struct(dst, w, h)
    refs = (RP * len(ref_planes_list))(*[planes_struct(r, w, h) for r in ref_planes_list])
    prep = np.zeros(frame.prep_elems, np.int16)
    prep[::2048] = 0                                   # touch the pages now, not inside the timed replay
    coef = frame.coef.copy()
    mask = np.zeros(16, np.uint8)
    t0 = time.perf_counter()
    if threads > 1:
        rc = rl.dav1d_replay_recon_mt(entry, bpc, C.byref(drp), refs, frame.mc.ctypes.data, len(frame.mc), frame.comp.ctypes.data,
                                      len(frame.comp), frame.itx.ctypes.data, len(frame.itx), prep.ctypes.data, coef.ctypes.data, threads)
        assert rc > 0, rc
        if timing is not None:
            timing["seconds"] = time.perf_counter() - t0
            timing["threads"] = rc
        return dst, prep, coef
    assert rl.dav1d_replay_mc(entry, bpc, C.byref(drp), refs, frame.mc.ctypes.data, len(frame.mc), prep.ctypes.data) == 0
    assert rl.dav1d_replay_comp(entry, bpc, C.byref(drp), frame.comp.ctypes.data, len(frame.comp),
                                prep.ctypes.data, mask.ctypes.data) == 0
    assert rl.dav1d_replay_itx(entry, bpc, C.byref(drp), frame.itx.ctypes.data, len(frame.itx), coef.ctypes.data) == 0
    if timing is not None:
        timing["seconds"] = time.perf_counter() - t0
        timing["threads"] = 1
    return dst, prep, coef


def hip_frame(ctx, frame, dst_planes, ref_planes_list, fused=False, packed=False, recon=False, twin_out=False):
    w, h, bpc = frame.w, frame.h, frame.bpc
    dst = ctx.picture(w, h, api.LAYOUT_I420, bpc)
    for pl in range(3):
        dst.upload(pl, dst_planes[pl])
    refs = []
    for rp in ref_planes_list:
        r = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            r.upload(pl, rp[pl])
        refs.append(r)
    prep = ctx.buffer(frame.prep_elems * 2)
    prep.zero()
    itx_tasks, coef_host = synth.pack_frame_coefs(frame) if packed else (frame.itx, frame.coef)
    coef = ctx.buffer_from(coef_host)
    if recon:
        rl = ctx.recon_list(dst, frame.mc, frame.comp, itx_tasks)
        if twin_out:
            rl.run_twin(dst, refs, prep, coef)
            assert dst.pic.twin_ok
            hip_frame.last_twin = [read_twin(ctx, dst, pl) for pl in range(3)]
        else:
            rl.run(dst, refs, prep, coef)
        rl.destroy()
    elif fused:
        il = ctx.inter_list(frame.mc, frame.comp)
        assert il.n_fused == len(frame.comp)        # the synthetic frames only hold avg compounds
        ctx.run_inter_list(il, dst, refs, prep)
        il.destroy()
    else:
        ctx.mc_batch(dst, refs, frame.mc, prep)
        if len(frame.comp):
            ctx.comp_batch(dst, frame.comp, prep, None)
    if not recon:
        ctx.itx_add_batch(dst, itx_tasks, coef)
    out = [dst.download(pl) for pl in range(3)]
    oprep = prep.download(np.int16, frame.prep_elems)
    ocoef = coef.download(coef_host.dtype, len(coef_host))
    if packed:
        assert np.array_equal(ocoef, coef_host), "a packed arena is read-only"
        ocoef = np.zeros_like(frame.coef)
    for o in [dst, prep, coef] + refs:
        o.free()
    return out, oprep, ocoef


def read_twin(ctx, pic, plane):
    """the tiled twin of a plane back in raster order (rows x stride), for comparisons"""
    stride = pic.stride_px(plane)
    rows = pic.padded_shape(plane)[0]
    bps = np.dtype(pic.dtype).itemsize
    ctx.sync()
    raw = np.zeros(stride * ((pic.pic.p[plane].h + 7) & ~7), pic.dtype)
    assert ctx.lib.dav1d_hip_download(ctx.h, raw.ctypes.data, pic.pic.twin[plane], raw.nbytes) == 0
    t = raw.reshape(-1, stride // 8, 8, 8)             # [tile row][tile column][row in tile][column in tile]
    return t.transpose(0, 2, 1, 3).reshape(-1, stride)[:rows]


@pytest.mark.parametrize("refs_tiled", [True, False], ids=["launches-write-the-twin", "retile-after"])
@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_recon_list_leaves_the_tiled_twin_of_its_picture(ctx, bpc, refs_tiled):
    """dav1d_hip_recon_list_run_twin: the picture of a frame without in-loop filters as later frames will read it.  With tiled
    references the paired, prediction and residual launches write the twin themselves (wide row pieces through tile_write_out, strips
    from the prediction kernels); with raster references a retile pass follows.  Raster planes equal the oracle's, and the twin,
    untiled, equals the raster planes over the visible area."""
    ctx.auto_retile = refs_tiled
    try:
        w, h = (512, 128) if ctx.backend == "emu" else (1280, 1024)
        frame = synth.make_frame(w, h, bpc, seed=915 + bpc, edge_frac=0.1)
        rng = np.random.default_rng(19 + bpc)
        refs = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
        dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
        want, _, want_coef = oracle_frame(util.default_oracle(), frame, dst0, refs)
        got, _, got_coef = hip_frame(ctx, frame, dst0, refs, recon=True, twin_out=True)
    finally:
        ctx.auto_retile = False
    assert np.array_equal(got_coef, want_coef)
    for pl in range(3):
        assert np.array_equal(got[pl], want[pl]), "plane %d differs from the oracle" % pl
        vh, vw = (h, w) if pl == 0 else (h // 2, w // 2)
        tw = hip_frame.last_twin[pl]
        bad = np.argwhere(tw[:vh, :vw] != want[pl][:vh, :vw])
        assert not len(bad), "twin of plane %d differs at %s (%d px)" % (pl, bad[0], len(bad))


@pytest.mark.parametrize("fused", [False, True], ids=["twostep", "fused"])
@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_frame_itx_mc_matches_oracle(ctx, bpc, fused, twin_refs):
    w, h = (1024, 64) if ctx.backend == "emu" else (1024, 576)   # 1024: exercises the +64 B stride rule
    frame = synth.make_frame(w, h, bpc, seed=31 + bpc, edge_frac=0.15)
    rng = np.random.default_rng(3 + bpc)
    refs = [synth.make_planes(rng, w, h, bpc, smooth=(i != 1)) for i in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    want, want_prep, want_coef = oracle_frame(util.default_oracle(), frame, dst0, refs)
    got, got_prep, got_coef = hip_frame(ctx, frame, dst0, refs, fused)
    for pl in range(3):
        bad = np.argwhere(got[pl] != want[pl])
        assert not len(bad), "plane %d differs at %s (%d px)" % (pl, bad[0], len(bad))
    if not fused:       # with fusion the prep arena is scratch: the int16 intermediates stay in registers
        assert np.array_equal(got_prep, want_prep)
    assert np.array_equal(got_coef, want_coef)
    assert not want_coef.any(), "all consumed coefficient slabs end up zeroed"


def test_threaded_replay_equals_serial_replay():
    """The all-cores leg of the CPU baseline must produce the pictures of the one-thread replay."""
    oracle = util.default_oracle()
    frame = synth.make_frame(512, 256, 10, seed=77)
    rng = np.random.default_rng(5)
    refs = [synth.make_planes(rng, 512, 256, 10) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, 512, 256, 10, smooth=False)
    one = oracle_frame(oracle, frame, dst0, refs)
    many = oracle_frame(oracle, frame, dst0, refs, threads=4)
    for pl in range(3):
        assert np.array_equal(one[0][pl], many[0][pl]), pl
    assert np.array_equal(one[1], many[1]) and np.array_equal(one[2], many[2])


@pytest.mark.parametrize("bpc", [8, 10])
def test_frame_from_packed_coefficients(ctx, bpc):
    """The sparse coefficient wire format (DAV1D_HIP_ITX_PACKED) over a whole frame: same pictures as the dense arena."""
    oracle = util.default_oracle()
    w, h = (256, 128) if ctx.backend == "emu" else (1024, 512)
    frame = synth.make_frame(w, h, bpc, seed=404 + bpc)
    rng = np.random.default_rng(1)
    refs = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    want, _, _ = oracle_frame(oracle, frame, dst0, refs)
    got, _, _ = hip_frame(ctx, frame, dst0, refs, fused=True, packed=True)
    for pl in range(3):
        assert np.array_equal(got[pl], want[pl]), pl


@pytest.mark.parametrize("fuse", ["0", "31", "6", "31-one-wave"], ids=["two-kernels", "all-paired", "default-paired", "all-paired-one-wave"])
@pytest.mark.parametrize("pipeline", ["0", "-1"], ids=["pipelined", "sequential"])
@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_recon_list_matches_oracle(ctx, bpc, pipeline, fuse, monkeypatch, twin_refs):
    """dav1d_hip_recon_list_*: predictions and residuals as one list, the residual launch of a transform size waiting only
    for the prediction launches under its blocks (two streams) — and the same list run strictly one phase after the other."""
    ctx.set_option("recon_pipeline", pipeline)
    if fuse.endswith("-one-wave"):          # the paired kernels as one wave per group (what large frames get) instead of the
        fuse = fuse.split("-")[0]           # cooperative form frames of this size get by default
        ctx.set_option("recon_coop_below", 0)
    ctx.set_option("recon_fuse", fuse)      # paired: prediction + residual of a block in one wave (recon.hip)
    w, h = (512, 128) if ctx.backend == "emu" else (1280, 1024)
    frame = synth.make_frame(w, h, bpc, seed=515 + bpc, edge_frac=0.1)
    rng = np.random.default_rng(9 + bpc)
    refs = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    want, _, want_coef = oracle_frame(util.default_oracle(), frame, dst0, refs)
    got, _, got_coef = hip_frame(ctx, frame, dst0, refs, recon=True)
    for pl in range(3):
        bad = np.argwhere(got[pl] != want[pl])
        assert not len(bad), "plane %d differs at %s (%d px)" % (pl, bad[0], len(bad))
    assert np.array_equal(got_coef, want_coef)


@pytest.mark.parametrize("pipeline", ["0", "-1"], ids=["pipelined", "sequential"])
def test_recon_list_with_transforms_smaller_than_their_prediction(ctx, pipeline, monkeypatch):
    """Blocks whose residual is coded as four transforms of half the size cannot be paired with their prediction: they
    take the prediction launch + residual launch route, where the 8x8 residual launch has to wait for the 16x16 prediction
    launch — next to paired blocks of the same frame."""
    import copy
    ctx.set_option("recon_pipeline", pipeline)
    bpc = 10
    w, h = (512, 128) if ctx.backend == "emu" else (1280, 1024)
    base = synth.make_frame(w, h, bpc, seed=808, edge_frac=0.05)
    rng = np.random.default_rng(81)
    stride = synth.plane_geometry(w, h, bpc, 1)[0][0]
    pick = np.flatnonzero((base.itx["tx"] == 2) & (base.itx["plane"] == 0))[::2]       # every other 16x16 luma transform
    cf, eob = synth.gen_coefs(rng, 1, 4 * len(pick), bpc)
    small = np.zeros(4 * len(pick), base.itx.dtype)
    off0 = len(base.coef)
    for q, (dy, dx) in enumerate(((0, 0), (0, 8), (8, 0), (8, 8))):
        part = small[q::4]
        part["dst_off"] = base.itx["dst_off"][pick] + dy * stride + dx
        part["cf_off"] = off0 + (np.arange(len(pick)) * 4 + q) * 64
        part["eob"] = eob[q::4]
        part["tx"], part["plane"] = 1, 0
    order = np.argsort(np.arange(4 * len(pick)).reshape(4, -1).T.ravel(), kind="stable")   # [block][quarter] coefficient order
    frame = copy.copy(base)
    frame.itx = np.concatenate([np.delete(base.itx, pick), small])
    slabs = np.zeros((4 * len(pick), 64), base.coef.dtype)
    for q in range(4):
        slabs[np.arange(len(pick)) * 4 + q] = cf[q::4]
    frame.coef = np.concatenate([base.coef, slabs.ravel()])
    del order
    refs = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    want, _, want_coef = oracle_frame(util.default_oracle(), frame, dst0, refs)
    got, _, got_coef = hip_frame(ctx, frame, dst0, refs, recon=True)
    for pl in range(3):
        bad = np.argwhere(got[pl] != want[pl])
        assert not len(bad), "plane %d differs at %s (%d px)" % (pl, bad[0], len(bad))
    assert np.array_equal(got_coef, want_coef)


@pytest.mark.parametrize("size", [(200, 136), (72, 40), (1000, 72)], ids=["200x136", "72x40", "1000x72"])
def test_recon_list_on_pictures_that_are_not_multiples_of_the_block_sizes(ctx, size, twin_refs):
    """Visible sizes that cut blocks (they still lie inside the padded planes, as in the reference's allocation)."""
    w, h = size
    bpc = 10
    frame = synth.make_frame(w, h, bpc, seed=w * 7 + h, edge_frac=0.2, mv_range_px=40)
    rng = np.random.default_rng(w + h)
    refs = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    want, _, want_coef = oracle_frame(util.default_oracle(), frame, dst0, refs)
    got, _, got_coef = hip_frame(ctx, frame, dst0, refs, recon=True)
    for pl in range(3):
        bad = np.argwhere(got[pl] != want[pl])
        assert not len(bad), "plane %d differs at %s (%d px)" % (pl, bad[0], len(bad))
    assert np.array_equal(got_coef, want_coef)


def test_recon_list_with_nothing_or_only_one_side(ctx):
    """Empty lists, predictions without residuals, residuals without predictions."""
    bpc, w, h = 8, 256, 128
    frame = synth.make_frame(w, h, bpc, seed=12)
    rng = np.random.default_rng(3)
    refs_h = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    oracle = util.default_oracle()
    import copy
    for mode in ("empty", "mc-only", "itx-only"):
        f = copy.copy(frame)
        if mode != "mc-only":
            f.itx = frame.itx if mode == "itx-only" else frame.itx[:0]
        else:
            f.itx = frame.itx[:0]
        if mode != "mc-only":
            f.mc, f.comp = frame.mc[:0], frame.comp[:0]
        want, _, _ = oracle_frame(oracle, f, dst0, refs_h)
        got, _, _ = hip_frame(ctx, f, dst0, refs_h, recon=True)
        for pl in range(3):
            assert np.array_equal(got[pl], want[pl]), (mode, pl)


def test_recon_list_refuses_another_geometry(ctx):
    frame = synth.make_frame(256, 128, 8, seed=2)
    a = ctx.picture(256, 128, api.LAYOUT_I420, 8)
    b = ctx.picture(1024, 128, api.LAYOUT_I420, 8)
    rl = ctx.recon_list(a, frame.mc, frame.comp, frame.itx)
    coef = ctx.buffer_from(frame.coef)
    prep = ctx.buffer(max(frame.prep_elems, 8) * 2)
    with pytest.raises(api.HipError):
        rl.run(b, [b] * frame.n_refs, prep, coef)
    rl.destroy()
    for o in (a, b, coef, prep):
        o.free()


@pytest.mark.parametrize("mode", ["1", "2"], ids=["all-shapes", "wide-shapes"])
def test_shapes_sharing_one_launch(ctx, mode):
    """DAV1D_HIP_MC_FUSED=1 / 2 (every tile shape / the shapes at least 16 wide in one source-ordered launch) must give
    the same pictures: the frame tests again, in a child process that has the switch set before the library reads it."""
    import os
    import subprocess
    import sys
    if os.environ.get("DAV1D_HIP_MC_FUSED"):
        pytest.skip("already inside the child run")
    env = dict(os.environ, DAV1D_HIP_MC_FUSED=mode)
    sel = "emu" if ctx.backend == "emu" else "hip"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", sel + " and fused", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("size", [(512, 128), (200, 136), (1000, 72)], ids=["512x128", "200x136", "1000x72"])
@pytest.mark.parametrize("bpc", [8, 10])
def test_recon_list_with_its_picture_in_the_twin_only(ctx, bpc, size):
    """dav1d_hip_recon_list_run_tiled: the frame's pixels go to the tiled twin and NOWHERE else (DAV1D_HIP_TWIN_ONLY: the raster planes
    keep what they held), a chain of two frames predicts the second from the first through that twin, and every way out gives the
    oracle's raster rows: plane download, dav1d_hip_host_picture_fetch in bands of rows, dav1d_hip_picture_untile.  Sizes that cut blocks
    reconstruct below the visible rows (the twin has the allocator's padding rows, like the raster planes)."""
    w, h = size
    if ctx.backend != "emu" and size == (512, 128):
        w, h = 1280, 1024
    oracle = util.default_oracle()
    frame = synth.make_frame(w, h, bpc, seed=w + 3 * h + bpc, edge_frac=0.1, mv_range_px=40)
    rng = np.random.default_rng(w + bpc)
    refs_h = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    want1, _, _ = oracle_frame(oracle, frame, dst0, refs_h)
    want2, _, _ = oracle_frame(oracle, frame, dst0, [want1] + refs_h[1:])         # the second frame predicts from the first
    refs = []
    for rp in refs_h:
        r = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            r.upload(pl, rp[pl])
        r.retile()
        refs.append(r)
    pics = []
    for _ in range(2):
        d = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            d.upload(pl, dst0[pl])
        pics.append(d)
    prep = ctx.buffer(frame.prep_elems * 2)
    prep.zero()
    rl = ctx.recon_list(pics[0], frame.mc, frame.comp, frame.itx)
    coef = ctx.buffer_from(frame.coef)
    rl.run_tiled(pics[0], refs, prep, coef)
    assert pics[0].pic.twin_ok == api.TWIN_ONLY
    # nothing reached the raster planes: they still hold what was uploaded
    raw = np.zeros((pics[0].padded_shape(0)[0], pics[0].stride_px(0)), pics[0].dtype)
    ctx.sync()
    assert ctx.lib.dav1d_hip_download(ctx.h, raw.ctypes.data, pics[0].pic.p[0].data, raw.nbytes) == 0
    assert np.array_equal(raw[:, :dst0[0].shape[1]], dst0[0]), "the raster planes were written"
    for pl in range(3):
        assert np.array_equal(pics[0].download(pl), want1[pl]), ("download un-tiles", pl)
    assert pics[0].pic.twin_ok == api.TWIN_ONLY
    # rows out to the host in bands (what a row-progress listener does)
    host = api.HostPictureBuf(ctx, w, h, api.LAYOUT_I420, bpc)
    edges = sorted({0, min(h, 56), min(h, 120), h})
    for r0, r1 in zip(edges[:-1], edges[1:]):
        host.fetch(pics[0].pic, r0, r1)
    host.wait()
    for pl in range(3):
        vh, vw = (h, w) if pl == 0 else ((h + 1) // 2, (w + 1) // 2)
        assert np.array_equal(host.plane(pl)[:vh, :vw], want1[pl][:vh, :vw]), ("fetch in bands", pl)
    host.release()
    # the second frame: reference 0 = the first frame's picture, read through the twin it lives in
    coef2 = ctx.buffer_from(frame.coef)
    rl.run_tiled(pics[1], [pics[0]] + refs[1:], prep, coef2)
    assert pics[1].pic.twin_ok == api.TWIN_ONLY
    pics[1].untile()
    assert pics[1].pic.twin_ok == 1
    ctx.sync()
    for pl in range(3):
        assert np.array_equal(pics[1].download(pl), want2[pl]), ("second frame of the chain, after untile", pl)
    rl.destroy()
    for o in pics + refs + [prep, coef, coef2]:
        o.free()


@pytest.mark.parametrize("bpc", [8, 10])
def test_raster_readers_of_a_picture_that_lives_in_its_twin_only(ctx, bpc):
    """A frame reconstructed in the tiled layout leaves its picture DAV1D_HIP_TWIN_ONLY: the raster planes are stale.  Launches that read
    raster planes of their references — warped (global / local motion) and scaled prediction — must see the frame all the same (they
    un-tile the picture first), not the stale planes: the second frame of a chain whose first frame was plain inter."""
    if util.ref_lib() is None:
        pytest.skip("needs the reference build (warp8x8)")
    oracle = util.Oracle("ref")
    w, h = 256, 128
    frame = synth.make_frame(w, h, bpc, seed=77 + bpc, edge_frac=0.0, mv_range_px=24)
    rng = np.random.default_rng(5 + bpc)
    refs_h = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst0 = synth.make_planes(rng, w, h, bpc, smooth=False)
    want1, _, _ = oracle_frame(util.default_oracle(), frame, dst0, refs_h)
    refs = []
    for rp in refs_h:
        r = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            r.upload(pl, rp[pl])
        r.retile()
        refs.append(r)
    first = ctx.picture(w, h, api.LAYOUT_I420, bpc)
    for pl in range(3):
        first.upload(pl, dst0[pl])          # what the raster planes hold while the picture lives in its twin
    prep = ctx.buffer(max(frame.prep_elems, 64 * 64) * 2)
    prep.zero()
    rl = ctx.recon_list(first, frame.mc, frame.comp, frame.itx)
    rl.run_tiled(first, refs, prep, ctx.buffer_from(frame.coef))
    assert first.pic.twin_ok == api.TWIN_ONLY
    # second frame: 8x8 warped blocks predicted from the first frame's luma
    pd = util.pix_dtype(bpc)
    dst = ctx.picture(64, 64, api.LAYOUT_I400, bpc)
    dplane = rng.integers(0, 1 << bpc, size=dst.padded_shape(0)).astype(pd)
    dst.upload(0, dplane)
    want = dplane.copy()
    sp = dst.stride_px(0)
    src = np.ascontiguousarray(want1[0])
    bps = pd().itemsize
    n = 32
    tasks = np.zeros(n, api.WARP_TASK)
    for i in range(n):
        bx, by = i % 8, i // 8
        dx, dy = int(rng.integers(4, w - 16)), int(rng.integers(4, h - 16))
        mx, my = (int(rng.integers(0, 0x2000)) - 0xa00 for _ in range(2))
        abcd = (rng.integers(0, 0x2000, size=4) - 0xa00).astype(np.int16)
        blk = want[by * 8:, bx * 8:]
        oracle.call(bpc, "warp8x8", 0, 0, blk.ctypes.data, want.strides[0], src.ctypes.data + dy * src.strides[0] + dx * bps, src.strides[0], abcd, mx, my)
        tasks[i] = (by * 8 * sp + bx * 8, dx, dy, mx, my, abcd, 64, 0, 0, 0, (0, 0, 0))
    ctx.warp_batch(dst, [first], tasks, prep)
    assert np.array_equal(dst.download(0), want), "the warped prediction did not read the first frame's pixels"
    rl.destroy()
