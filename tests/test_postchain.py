"""The in-loop filter chain of a whole frame -- deblocking, CDEF, loop restoration, film grain -- on the task lists
tests/synth_frames.py derives from the frame's transform grid: HIP backend through the C ABI vs the oracle's DSP entries
driven by oracle/replay.c the way the reference drivers drive them (src/lf_apply_tmpl.c, src/cdef_apply_tmpl.c,
src/lr_apply_tmpl.c), stage by stage on identical inputs."""
import ctypes as C

import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth
from test_frame import RP, planes_struct


def oracle_post(oracle, post, recon, w, h, bpc, with_grain=True):
    """recon planes -> (deblocked, cdef, restored, grain) planes on the host."""
    rl = util.replay_lib()
    entry = C.cast(oracle._entry, C.c_void_p)
    lut = np.zeros(144, np.uint8)
    lut[:64], lut[64:128] = post.lut_e, post.lut_i
    d = synth.copy_planes(recon)
    assert rl.dav1d_replay_lf(entry, bpc, C.byref(planes_struct(d, w, h)), post.lf.ctypes.data, len(post.lf), post.lvl.ctypes.data,
                              post.b4_stride, lut.ctypes.data) == 0
    c = synth.copy_planes(d)
    assert rl.dav1d_replay_cdef(entry, bpc, 1, C.byref(planes_struct(d, w, h)), C.byref(planes_struct(c, w, h)), post.cdef.ctypes.data,
                                len(post.cdef), post.cdef_damping) == 0
    r = synth.copy_planes(c)
    assert rl.dav1d_replay_lr(entry, bpc, C.byref(planes_struct(c, w, h)), C.byref(planes_struct(d, w, h)), C.byref(planes_struct(r, w, h)),
                              post.lr.ctypes.data, len(post.lr)) == 0
    g = None
    if with_grain:
        from test_filmgrain import fg_driver
        g = synth.copy_planes(r)
        src = synth.copy_planes(r)      # dav1d_apply_grain pads the luma of its input by one pixel (src/fg_apply_tmpl.c:193-199)
        sp = (C.c_void_p * 3)(*[p.ctypes.data for p in src])
        gp = (C.c_void_p * 3)(*[p.ctypes.data for p in g])
        assert fg_driver(oracle).apply_grain(bpc, C.addressof(post.fg), w, h, 1, 0, gp, sp, g[0].strides[0], g[1].strides[0]) == 0
    return d, c, r, g


def hip_post(ctx, post, recon, w, h, bpc, with_grain=True):
    pics = [ctx.picture(w, h, api.LAYOUT_I420, bpc) for _ in range(5)]
    rec, dbl, cdf, res, grn = pics
    for pl in range(3):
        dbl.upload(pl, recon[pl])
    lvl = ctx.buffer_from(post.lvl)
    ctx.lf_batch(dbl, post.lf, lvl, post.b4_stride, post.lut_e, post.lut_i)
    d = [dbl.download(pl) for pl in range(3)]
    for pl in range(3):
        cdf.upload(pl, d[pl])
    ctx.cdef_batch(cdf, dbl, post.cdef, post.cdef_damping)
    c = [cdf.download(pl) for pl in range(3)]
    for pl in range(3):
        res.upload(pl, c[pl])
    ctx.lr_batch(res, cdf, dbl, post.lr)
    r = [res.download(pl) for pl in range(3)]
    g = None
    if with_grain:
        ctx.fg_apply(grn, res, post.fg)
        g = [grn.download(pl) for pl in range(3)]
    for o in pics + [lvl]:
        o.free()
    return d, c, r, g


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_post_filter_chain_matches_oracle(ctx, bpc):
    oracle = util.default_oracle()
    w, h = (256, 192) if ctx.backend == "emu" else (1024, 576)
    frame = synth.make_frame(w, h, bpc, seed=77 + bpc)
    post = synth.make_post_filters(frame, seed=5 + bpc)
    rng = np.random.default_rng(bpc)
    recon = synth.make_planes(rng, w, h, bpc, smooth=True)
    want = oracle_post(oracle, post, recon, w, h, bpc)
    got = hip_post(ctx, post, recon, w, h, bpc)
    for stage, a, b in zip(("deblock", "cdef", "restoration", "grain"), got, want):
        if b is None:
            continue
        for pl in range(3):
            vh, vw = (h, w) if pl == 0 else (h // 2, w // 2)
            bad = np.argwhere(a[pl][:vh, :vw] != b[pl][:vh, :vw])
            assert not len(bad), "%s plane %d differs at %s (%d px)" % (stage, pl, bad[0], len(bad))
    assert any(np.any(got[0][pl] != recon[pl]) for pl in range(3)), "deblocking must change something"


def oracle_intra(oracle, ip, planes, w, h, bpc):
    """The intra pass on host planes (in place): per wave, prediction of every block through the reference's
    dav1d_prepare_intra_edges + intra_pred entries, then the residuals through oracle/replay.c."""
    import test_ipred
    rl = util.replay_lib()
    entry = C.cast(oracle._entry, C.c_void_p)
    coef = ip.coef.copy()
    rp = planes_struct(planes, w, h)
    for pred, itx in ip.batches:
        for k in range(len(pred)):
            test_ipred.oracle_task(oracle, bpc, planes, pred[k], 1, None)
        assert rl.dav1d_replay_itx(entry, bpc, C.byref(rp), itx.ctypes.data, len(itx), coef.ctypes.data) == 0
    return coef


def hip_intra(ctx, ip, pic, timed=False, graph=False, paired=True, flow=False, sb=False):
    """Runs the intra pass from device-resident lists, every wave enqueued back to back on the context's stream
    (prediction of wave k, residuals of wave k, prediction of wave k + 1, ...), no host round trip in between.
    graph: record the whole chain once (dav1d_hip_graph_*) and replay it as one HIP graph.
    paired: through dav1d_hip_intra_list_* (4x4 / 8x8 blocks: prediction + residual in one wave) instead of the prediction
    list + one residual list per step.
    flow: through dav1d_hip_intra_flow_* — the whole pass as one launch, steps handed over between running waves.
    sb: through dav1d_hip_intra_sb_* — a workgroup per superblock, a launch per level of superblocks.
    timed (bench.py only; needs torch for the events): returns the device time of the whole pass in ms, else 0."""
    import ctypes as C
    coef = ctx.buffer_from(ip.coef)
    lib = ctx.lib
    fl = None
    sl = None
    if sb:
        sl = ctx.intra_sb(ip.batches, pic)
        hip_intra.sb_levels, hip_intra.sb_superblocks = sl.n_levels, sl.n_superblocks
        xl, plist, ilists = None, None, []
    elif flow:
        fl = ctx.intra_flow(ip.batches)
        xl, plist, ilists = None, None, []
    elif paired:
        xl = ctx.intra_list(ip.batches)
        plist, ilists = None, []
    else:
        xl = None
        plist = ctx.ipred_list([b[0] for b in ip.batches])
        ilists = [ctx.itx_list(b[1]) for b in ip.batches]
    lib.dav1d_hip_sync(ctx.h)

    def chain():
        if sl is not None:
            sl.run(pic, coef)
            return
        if fl is not None:
            fl.run(pic, coef)
            return
        if xl is not None and not graph:
            xl.run_all(pic, coef)              # dav1d_hip_intra_list_run_all: the batches in order, one call
            return
        for k in range(len(ip.batches)):
            if xl is not None:
                xl.run_batch(k, pic, coef)
            else:
                plist.run_batch(k, pic)
                ctx.run_itx_list(ilists[k], pic, coef)

    g = None
    if graph:
        ctx.graph_begin()
        chain()
        g = ctx.graph_end()
        hip_intra.last_nodes = int(lib.dav1d_hip_graph_nodes(g))
    ev = None
    if timed:
        import torch
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        stream = torch.cuda.ExternalStream(lib.dav1d_hip_stream(ctx.h))
        ev[0].record(stream)
    if g is not None:
        ctx.graph_launch(g)
    else:
        chain()
    ms = 0.0
    if ev:
        ev[1].record(stream)
        lib.dav1d_hip_sync(ctx.h)
        ms = ev[0].elapsed_time(ev[1])
    lib.dav1d_hip_sync(ctx.h)
    left = coef.download(ip.coef.dtype, len(ip.coef))
    if g is not None:
        ctx.graph_destroy(g)
    if sl is not None:
        sl.status()                            # no superblock of the one-launch form gave up waiting for a neighbour
        sl.destroy()
    elif fl is not None:
        hip_intra.flow_status = fl.status()
        fl.destroy()
        assert hip_intra.flow_status == (hip_intra.flow_status[0], fl.n_units, 0), hip_intra.flow_status
    elif xl is not None:
        xl.destroy()
    else:
        plist.destroy()
    for l in ilists:
        l.destroy()
    coef.free()
    assert not left.any(), "every coefficient slab must come back zeroed"
    return ms


@pytest.mark.parametrize("paired", [True, False, "flow", "sb", "sb-lds"], ids=["paired", "two-launches", "one-launch", "superblocks", "superblocks-lds"])
@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_intra_wavefront_pass_matches_oracle(ctx, bpc, paired):
    oracle = util.default_oracle()
    w, h = (256, 192) if ctx.backend == "emu" else (1024, 576)
    frame = synth.make_frame(w, h, bpc, seed=91 + bpc)
    ip = synth.make_intra_pass(frame, seed=17 + bpc)
    assert len(ip.batches) >= 1 and ip.n_blocks >= 3
    rng = np.random.default_rng(bpc)
    planes = synth.make_planes(rng, w, h, bpc, smooth=True)
    pic = ctx.picture(w, h, api.LAYOUT_I420, bpc)
    for pl in range(3):
        pic.upload(pl, planes[pl])
    want = synth.copy_planes(planes)
    oracle_intra(oracle, ip, want, w, h, bpc)
    if paired == "sb-lds":
        ctx.set_option("intra_sb_lds", 1)
    try:
        hip_intra(ctx, ip, pic, paired=paired is True, flow=paired == "flow", sb=paired in ("sb", "sb-lds"))
    finally:
        ctx.set_option("intra_sb_lds", 0)
    for pl in range(3):
        vh, vw = (h, w) if pl == 0 else (h // 2, w // 2)
        bad = np.argwhere(pic.download(pl)[:vh, :vw] != want[pl][:vh, :vw])
        assert not len(bad), "plane %d differs at %s (%d px)" % (pl, bad[0], len(bad))
    assert any(np.any(want[pl] != planes[pl]) for pl in range(3))
    pic.free()


@pytest.mark.gpu
def test_intra_wavefront_pass_replayed_as_a_graph():
    """The same chain recorded once and replayed through dav1d_hip_graph_*: identical picture."""
    ctx = util.make_context("hip")
    oracle = util.default_oracle()
    bpc, w, h = 10, 1024, 576
    frame = synth.make_frame(w, h, bpc, seed=191)
    ip = synth.make_intra_pass(frame, seed=117)
    rng = np.random.default_rng(4)
    planes = synth.make_planes(rng, w, h, bpc, smooth=True)
    pic = ctx.picture(w, h, api.LAYOUT_I420, bpc)
    for pl in range(3):
        pic.upload(pl, planes[pl])
    want = synth.copy_planes(planes)
    oracle_intra(oracle, ip, want, w, h, bpc)
    hip_intra(ctx, ip, pic, graph=True)
    assert hip_intra.last_nodes >= len(ip.batches)
    for pl in range(3):
        vh, vw = (h, w) if pl == 0 else (h // 2, w // 2)
        assert np.array_equal(pic.download(pl)[:vh, :vw], want[pl][:vh, :vw]), pl
    pic.free()
    ctx.close()


def test_graph_capture_is_refused_where_it_cannot_work():
    """The SIMT emulator runs launches on the spot: recording must fail loudly (-ENOSYS), not silently do nothing."""
    ctx = util.make_context("emu")
    with pytest.raises(api.HipError):
        ctx.graph_begin()
    ctx.close()
