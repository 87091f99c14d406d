"""The error path under dav1d's own loop, and Dav1dSettings.inloop_filters ("what's missing" 1 and 2).

Streams that are NOT repaired: a tile is cut short on purpose (its symbol decoder runs out of data: msac.cnt <= -15, reference
src/decode.c:2743, the tile task fails, src/thread_task.c:745-748), or its payload is re-rolled without checking that dav1d accepts
it.  Both ways of decoding — dav1d alone and with the binding (dav1d_amd/host/dav1d_glue.c) behind its pass 1 — must then agree on
everything an application sees: how many pictures come out, which ones (order hints) and their pixels, how many errors dav1d reports,
and nothing may hang or stay behind: the frame that fails in pass 1 (f->task_thread.error, dav1d_decode_frame_exit with its cf wipe,
src/decode.c:3242-3251), FRAME_ERROR in progress[1] for the frames that predict from it (check_tile, src/thread_task.c:393-439; the
binding's stage 2 for frames listed ahead of their references), the lister and frame objects of a frame that never reached the backend
(dav1d_hip_live_objects back to where it was)."""
import ctypes as C

import pytest

import util
import av1_obu
import stream_util as su
from dav1d_amd import _lib

pytestmark = pytest.mark.skipif(su.lib() is None, reason="needs oracle/_ref_hooked (the reference build with the hook patch)")

SIZES = [(256, 192), (200, 136), (320, 176), (448, 264)]


def hip_lib_path(ctx):
    return util.emu_lib_path() if ctx.backend == "emu" else _lib.DEFAULT_PATH


def live(path):
    lib = C.CDLL(path)
    out = (C.c_longlong * 4)()
    assert lib.dav1d_hip_live_objects(out) == 0
    return list(out)


def damaged_stream(seed, lib):
    """a valid stream (repaired), then one tile of one frame damaged: which frame / tile / how is drawn from the seed"""
    layout, bpc = (1, 3, 0, 2)[seed % 4], (8, 10, 12)[seed % 3]
    w, h = SIZES[(seed // 4) % len(SIZES)]
    n_frames = 7
    sw = av1_obu.make_stream(w, h, layout, bpc, n_frames, 7000 + seed, sb128=bool(seed & 8))
    su.repair(sw, lib, max_rounds=8000)
    frames = [i for i, u in enumerate(sw.units) if u["frame"] is not None]
    k = frames[(seed // 3) % len(frames)]                       # key frame, inter frames, hidden / intra-only frames: whatever the seed hits
    tiles = sw.units[k]["tiles"]
    ti = (seed // 5) % len(tiles)
    how = (0, 0, 1, 0, 2, 0)[(seed // 4) % 6]          # mostly tiles of a few bytes: those reliably fail (a tile cut in the middle often still decodes:
                                                # the symbol decoder pads with zeros and only a heavy overread is an error)
    if how == 0:
        sw.truncate(k, ti, 4 + seed % 13)                       # a tile of a few bytes: the symbol decoder overreads
    elif how == 1:
        sw.truncate(k, ti, max(8, tiles[ti][1] // (3 + seed % 5)))     # cut somewhere in the middle
    else:
        sw.reroll(k, ti, (seed * 37) % max(1, tiles[ti][1] // 2))      # other random bytes: dav1d may accept them (then it is just another stream) or not
    return sw, k, ti, ("tiny tile", "cut tile", "re-rolled tile")[how]


def check_seed(ctx, seed, n_devices=0):
    lib = hip_lib_path(ctx)
    before = live(lib)
    sw, k, ti, how = damaged_stream(seed, lib)
    units = [u["data"] for u in sw.units]
    want = su.decode(units, 0, lib, threads=2 + seed % 5, frame_delay=2 + seed % 3)
    got = su.decode(units, 1, lib, threads=2 + seed % 5, frame_delay=2 + seed % 3, free_listing=seed & 1, pack=not seed & 16, row_progress=(seed >> 5) & 1,
                    n_devices=n_devices)
    what = "seed %d (%s in unit %d tile %d)" % (seed, how, k, ti)
    assert got["rc"] == 0, "%s: the BACKEND reported a failure of its own (a frame dav1d rejects is not one)" % what
    assert got["errors"] == want["errors"], "%s: %d errors reported with the backend, %d by dav1d alone" % (what, got["errors"], want["errors"])
    assert [p[0] for p in got["pictures"]] == [p[0] for p in want["pictures"]], "%s: other pictures come out" % what
    diff = su.compare(want, got)
    assert diff is None, "%s: %s" % (what, diff)
    assert live(lib) == before, "%s: objects of the backend left behind: %s -> %s (contexts, frames, listers, host pictures)" % (what, before, live(lib))
    return want["errors"], len(want["pictures"])


@pytest.mark.parametrize("seed", range(1, 9))
def test_damaged_streams_fail_the_same_way_with_the_backend(ctx, seed):
    if ctx.backend != "emu":
        pytest.skip("the GPU run takes the sweep below")
    check_seed(ctx, seed)


@pytest.mark.parametrize("seed", range(1, 9))
def test_damaged_streams_over_two_devices_in_one_process(ctx, seed):
    """the same damaged streams with the binding's frames ending on two devices in turn: a frame that failed on one device is the failed
    reference of a frame on the other (FRAME_ERROR in progress[1], no mirror is made of it), nothing is left behind on either"""
    import ctypes as C
    if C.CDLL(hip_lib_path(ctx)).dav1d_hip_device_count() < 2:
        pytest.skip("one device here")
    check_seed(ctx, seed, n_devices=2)


@pytest.mark.gpu
@pytest.mark.parametrize("part", range(4))
def test_sweep_of_damaged_streams_on_the_gpu(part):
    """>= 50 damaged streams on the MI355X (DAV1D_ERROR_SEEDS overrides the count), in four parts; most of them must really have failed"""
    import os
    ctx = util.make_context("hip")
    ctx.backend = "hip"
    n = int(os.environ.get("DAV1D_ERROR_SEEDS", "56"))
    failed = 0
    try:
        for seed in range(100 + part, 100 + n, 4):
            errors, _ = check_seed(ctx, seed)
            failed += errors > 0
    finally:
        ctx.close()
    assert failed >= (n // 4) // 3, "only %d of %d damaged streams made dav1d report an error" % (failed, n // 4)


@pytest.mark.parametrize("off", [1, 2, 4, 3, 7], ids=["no-deblock", "no-cdef", "no-restoration", "no-deblock-no-cdef", "none"])
def test_inloop_filters_the_application_switched_off(ctx, off):
    """Dav1dSettings.inloop_filters (include/dav1d/dav1d.h:61-69; reference src/recon_tmpl.c:1988, 2014, 2027, 2089): `dav1d --inloopfilters`
    with the backend must give dav1d's pictures — which differ from the fully filtered ones."""
    lib = hip_lib_path(ctx)
    seeds = (2, 5) if ctx.backend == "emu" else (2, 5, 9, 14)
    for seed in seeds:
        layout, bpc = (1, 3, 0, 2)[seed % 4], (8, 10, 12)[seed % 3]
        w, h = SIZES[(seed // 4) % len(SIZES)]
        sw = av1_obu.make_stream(w, h, layout, bpc, 5, 9100 + seed, sb128=bool(seed & 8))
        su.repair(sw, lib, max_rounds=8000)
        units = [u["data"] for u in sw.units]
        full = su.decode(units, 0, lib)
        want = su.decode(units, 0, lib, filters_off=off)
        got = su.decode(units, 1, lib, filters_off=off)
        assert not want["errors"] and not got["errors"]
        diff = su.compare(want, got)
        assert diff is None, "seed %d, filters off %d: %s" % (seed, off, diff)
        assert su.compare(full, want) is not None, "seed %d: switching filters off (%d) changed nothing: the stream does not exercise it" % (seed, off)
