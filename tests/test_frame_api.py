"""Driver-level boundary (dav1d_hip_frame_*): tasks submitted tile-sbrow by tile-sbrow from several threads, one
dav1d_hip_frame_end() for the whole frame -- reconstruction, deblocking, CDEF, loop restoration, film grain -- against the
oracle running the same stages through oracle/replay.c."""
import errno
import threading

import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth
import test_frame
import test_postchain


@pytest.mark.parametrize("bpc", [8, 10, 12])
@pytest.mark.parametrize("bands", [8, 0, -1, -8, -100], ids=["banded", "staged", "staged-async-end", "banded-async-progress", "staged-async-progress"])
def test_frame_in_flight_matches_oracle(ctx, bpc, bands, monkeypatch):
    """bands: the post filters pipelined over superblock-row bands (three streams) / one stage after the other; -1: the latter
    through dav1d_hip_frame_end_async (the frame runs on a thread of the library, completion arrives through the callback and
    dav1d_hip_frame_progress — the hook for dav1d's progress publication, src/thread_task.c:888-896); -8: banded and
    asynchronous with the row-granular progress callback — every time rows are published they are copied out of the filtered
    picture to pinned host planes (dav1d_hip_host_picture_*, the buffers behind a Dav1dPicAllocator) while the bands below are
    still being filtered, and what was copied must be the final picture; -100: the same listener on the stage-by-stage schedule —
    only the LAST stage (restoration) runs in bands of 256 rows with an event behind each (frame_lr_banded, frame.hip)."""
    async_end = bands < 0
    last_stage_rows = bands == -100
    bands = 0 if bands in (-1, -100) else abs(bands)
    ctx.set_option("post_bands", bands)

    oracle = util.default_oracle()
    w, h = (64, 768) if ctx.backend == "emu" else (1024, 1152)
    frame = synth.make_frame(w, h, bpc, seed=31 + bpc, edge_frac=0.1)
    post = synth.make_post_filters(frame, seed=9 + bpc)
    rng = np.random.default_rng(5)
    ref_host = [synth.make_planes(rng, w, h, bpc) for _ in range(frame.n_refs)]
    dst_host = synth.make_planes(rng, w, h, bpc, smooth=False)
    want_rec, _, _ = test_frame.oracle_frame(oracle, frame, dst_host, ref_host)
    # some regions are then re-coded as intra blocks (wavefront steps), on top of the inter reconstruction
    ip = synth.make_intra_pass(frame, seed=77 + bpc)
    want_rec = synth.copy_planes(want_rec)
    test_postchain.oracle_intra(oracle, ip, want_rec, w, h, bpc)
    want = test_postchain.oracle_post(oracle, post, want_rec, w, h, bpc)

    cur = ctx.picture(w, h, api.LAYOUT_I420, bpc)
    grain = ctx.picture(w, h, api.LAYOUT_I420, bpc)
    refs = []
    for rp in ref_host:
        r = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        for pl in range(3):
            r.upload(pl, rp[pl])
        refs.append(r)
    for pl in range(3):
        cur.upload(pl, dst_host[pl])
    prep = ctx.buffer(frame.prep_elems * 2)
    prep.zero()
    coef = ctx.buffer_from(np.concatenate([frame.coef, ip.coef]))       # one arena: the intra residuals behind the inter ones
    lvl = ctx.buffer_from(post.lvl)

    f = ctx.frame(cur, refs)
    f.set_filters(lvl, post.b4_stride, post.lut_e, post.lut_i, post.cdef_damping, post.fg, 0)
    # "tile-sbrows": interleaved chunks of the lists, submitted from four threads; a compound task travels with the two
    # PREP tasks it consumes only if they sit in the same list, so the compound list goes in with the first chunk
    nchunk = 8
    mc_chunks = np.array_split(frame.mc, nchunk)
    itx_chunks = np.array_split(frame.itx, nchunk)
    errs = []

    def worker(k):
        try:
            f.submit_tile_sbrow(mc_chunks[k], frame.comp if k == 0 else frame.comp[:0], itx_chunks[k])
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(k,)) for k in range(nchunk)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in reversed(range(len(ip.batches))):                           # steps may arrive in any order
        pt, it = ip.batches[k]
        it = it.copy()
        it["cf_off"] += len(frame.coef)
        f.submit_intra_step(k, pt, it)
    # loop restoration units must keep raster order: one submission per plane and stripe row
    f.submit_filter_sbrow(post.lf, post.cdef, post.lr)
    host = None
    if async_end:
        seen, steps = [], []
        assert f.progress() == 0
        if bands or last_stage_rows:
            host = api.HostPictureBuf(ctx, w, h, api.LAYOUT_I420, bpc)

            def on_rows(rows, pic):
                host.fetch(pic, steps[-1] if steps else 0, rows)          # rows [previous, rows) are final: out they go
                steps.append(rows)
            f.set_progress_callback(on_rows)
        f.end_async(coef, prep, None, grain, done=lambda rc: seen.append((rc, f.progress())))
        filtered = f.wait()
        assert seen == [(0, h)] and f.progress() == h, seen
        if bands or last_stage_rows:
            nb = min(bands, (h + 255) // 256) if bands else (h + 255) // 256
            assert len(steps) == nb and steps == sorted(set(steps)) and steps[-1] == h, steps
            # restoration stripes start 8 rows above the 64-row grid: a band's last stripe ends 56 rows into the next band
            assert all(r % 256 == 56 for r in steps[:-1]), steps
            host.wait()
    else:
        filtered = f.end(coef, prep, None, grain)
    assert f.post_bands() == (min(bands, (h + 255) // 256) if bands else 0)
    out = api.DevicePicture.view(ctx, filtered, w, h, api.LAYOUT_I420, bpc)
    for pl in range(3):
        vh, vw = (h, w) if pl == 0 else (h // 2, w // 2)
        assert np.array_equal(cur.download(pl)[:vh, :vw], want[0][pl][:vh, :vw]), ("deblocked", pl)
        assert np.array_equal(out.download(pl)[:vh, :vw], want[2][pl][:vh, :vw]), ("restored", pl)
        if want[3] is not None:
            assert np.array_equal(grain.download(pl)[:vh, :vw], want[3][pl][:vh, :vw]), ("grain", pl)
        if host is not None:
            assert np.array_equal(host.plane(pl)[:vh, :vw], want[2][pl][:vh, :vw]), ("rows copied out as they were published", pl)
    if host is not None:
        assert host.hp.stride[0] == cur.pic.p[0].stride and host.hp.stride[1] == cur.pic.p[1].stride
        host.release()
    f.destroy()
    for o in [cur, grain, prep, coef, lvl] + refs:
        o.free()


def test_two_devices_in_one_process_pictures_belong_to_their_device(ctx):
    """include/dav1d_hip.h, "several devices in ONE process": a context belongs to its device, a frame's picture has to live there
    (-EXDEV otherwise), dav1d_hip_picture_copy_peer makes a picture resident on another device — planes and tiled twin.  Here on the two
    emulated devices (tests/conftest.py): allocations carry their device, a peer copy between the wrong ones fails."""
    import ctypes as C
    from dav1d_amd import api
    lib = ctx.lib
    if lib.dav1d_hip_device_count() < 2:
        pytest.skip("one device here")
    other = api.Context(1, lib_path=ctx.lib_path)
    try:
        assert lib.dav1d_hip_context_device(ctx.h) == 0 and lib.dav1d_hip_context_device(other.h) == 1
        w, h, bpc = 200, 120, 10
        rng = np.random.default_rng(77)
        assert lib.dav1d_hip_context_use(ctx.h) == 0
        a = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        planes = [rng.integers(0, 1 << bpc, a.padded_shape(pl), dtype=np.uint16) for pl in range(3)]
        for pl in range(3):
            a.upload(pl, planes[pl])
        a.retile()
        assert lib.dav1d_hip_context_use(other.h) == 0
        b = other.picture(w, h, api.LAYOUT_I420, bpc)
        assert lib.dav1d_hip_picture_twin_alloc(other.h, C.byref(b.pic)) == 0
        assert lib.dav1d_hip_picture_device(C.byref(a.pic)) == 0 and lib.dav1d_hip_picture_device(C.byref(b.pic)) == 1
        # a frame of device 1 cannot be begun on a picture of device 0, nor ended with such a reference
        fh = C.c_void_p()
        assert lib.dav1d_hip_frame_begin(other.h, C.byref(fh), C.byref(a.pic), None, 0) == -errno.EXDEV
        refs = (api.Picture * 1)(a.pic)
        assert lib.dav1d_hip_frame_begin(other.h, C.byref(fh), C.byref(b.pic), refs, 1) == 0
        assert lib.dav1d_hip_frame_end(fh, None, None, None, None, None) == -errno.EXDEV
        lib.dav1d_hip_frame_destroy(fh)
        # ... until it has been made resident there (direct copies where the devices are peers; the caller's device stays current)
        assert lib.dav1d_hip_enable_peer_access(other.h, ctx.h) in (0, 1, 2) and lib.dav1d_hip_current_device() == 1
        assert lib.dav1d_hip_enable_peer_access(ctx.h, ctx.h) == 0
        assert lib.dav1d_hip_picture_copy_peer(other.h, C.byref(b.pic), ctx.h, C.byref(a.pic)) == 0
        other.sync()
        assert b.pic.twin_ok == 1
        for pl in range(3):
            assert np.array_equal(b.download(pl), planes[pl])
        # the twin came along: un-tiling it gives the planes again
        b.pic.twin_ok = api.TWIN_ONLY
        for pl in range(3):
            vis = (h if not pl else (h + 1) // 2), (w if not pl else (w + 1) // 2)
            assert np.array_equal(b.download(pl)[:vis[0], :vis[1]], planes[pl][:vis[0], :vis[1]])
        b.pic.twin_ok = 1
        # geometry has to agree
        c = other.picture(w + 16, h, api.LAYOUT_I420, bpc)
        assert lib.dav1d_hip_picture_copy_peer(other.h, C.byref(c.pic), ctx.h, C.byref(a.pic)) == -errno.EINVAL
        c.free(); b.free()
        assert lib.dav1d_hip_context_use(ctx.h) == 0
        a.free()
    finally:
        lib.dav1d_hip_context_use(ctx.h)
        other.close()
        lib.dav1d_hip_context_use(ctx.h)
