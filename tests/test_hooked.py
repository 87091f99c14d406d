"""The backend inside dav1d's own task loop: dav1d_open / dav1d_submit_frame / dav1d_worker_task threads /
check_tile dependencies / dav1d_get_picture are the reference's, src/thread_task.c carries the hook points of INTEGRATION.md 2
(patches/dav1d-1.5.4-hip.patch), and a chain of frames — a key frame, then inter frames predicting from the three frames before
them — is decoded twice from the same injected pass-1 output: by the reference's own pass 2 + in-loop filters on its worker
threads, and by the glue of INTEGRATION.md (allocator on dav1d_hip_host_picture_*, dav1d_hip_lister_tile_sbrow /
_filter_sbrow from the tile and filter tasks, dav1d_hip_frame_end when the frame's tasks are through).  Every output picture must be
identical."""
import numpy as np
import pytest

import util
import hooked_util as hk
from dav1d_amd import _lib

pytestmark = pytest.mark.skipif(hk.lib() is None, reason="needs oracle/_ref_hooked (the reference build with the hook patch)")


def hip_lib_path(ctx):
    return util.emu_lib_path() if ctx.backend == "emu" else _lib.DEFAULT_PATH


CASES = [
    ("420_10_filters", 384, 256, 10, dict(tiles=(2, 1))),
    ("420_8_tiles_2x2_no_filters", 320, 200, 8, dict(tiles=(2, 2), filters=None)),
    ("444_12_sb64_dav1d_dependencies", 256, 136, 12, dict(layout=3, sb128=False, tiles=(1, 1), free_listing=0)),
    # rows published band by band from inside the frame's last stage (dav1d_hip_frame_set_progress_callback -> the store of
    # src/thread_task.c:888-896), and dav1d's own rule for when a tile task may start: check_tile() waits for the rows of the references
    ("420_10_rows_published_per_band_dav1d_dependencies", 320, 1100, 10, dict(tiles=(1, 1), free_listing=0, row_progress=1)),
]


@pytest.mark.parametrize("name,w,h,bpc,kw", CASES, ids=[c[0] for c in CASES])
def test_chain_through_the_task_loop_equals_dav1d(ctx, name, w, h, bpc, kw):
    n_frames = 6 if ctx.backend == "emu" else 9
    kw = dict(kw)
    want_s, n_fc, want = hk.run(hk.params(w, h, bpc, n_frames, mode=0, **kw), hip_lib_path(ctx))
    assert n_fc >= 3
    got_s, _, got = hk.run(hk.params(w, h, bpc, n_frames, mode=1, **kw), hip_lib_path(ctx))
    for k in range(n_frames):
        for pl in range(len(want[k])):
            bad = np.argwhere(want[k][pl] != got[k][pl])
            assert not len(bad), "frame %d plane %d differs at %s (%d pixels)" % (k, pl, bad[0], len(bad))
    if kw.get("row_progress"):
        assert hk.run.last_row_publications >= n_frames, hk.run.last_row_publications      # bands of 256 rows: several per frame
    # the chain is a chain: frames differ from each other, and an inter frame is not what the key frame was
    assert not np.array_equal(want[0][0], want[1][0]) and not np.array_equal(want[1][0], want[n_frames - 1][0])


def n_devices_here(ctx):
    import ctypes as C
    l = C.CDLL(hip_lib_path(ctx))
    return l.dav1d_hip_device_count()


@pytest.mark.parametrize("name,w,h,bpc,kw", CASES, ids=[c[0] for c in CASES])
def test_chain_over_two_devices_in_one_process_equals_dav1d(ctx, name, w, h, bpc, kw):
    """dav1d is ONE process (n_fc frame contexts, reference src/internal.h:354-388): Dav1dHipGlueOptions.n_devices = 2 ends the frames on two
    devices in turn, and a frame that predicts from pictures of the other device has them copied over first (dav1d_hip_picture_copy_peer).
    On this box: two emulated devices ($DAV1D_EMU_DEVICES, tests/conftest.py) whose allocations are tagged — a frame handed a picture of
    the wrong device ends with -EXDEV, a peer copy between the wrong devices fails."""
    if n_devices_here(ctx) < 2:
        pytest.skip("one device here")
    n_frames = (4 if h > 1000 else 6) if ctx.backend == "emu" else 9          # (the tall case is a minute of emulation at six frames)
    kw = dict(kw)
    _, _, want = hk.run(hk.params(w, h, bpc, n_frames, mode=0, **kw), hip_lib_path(ctx))
    _, _, got = hk.run(hk.params(w, h, bpc, n_frames, mode=1, n_devices=2, **kw), hip_lib_path(ctx))
    for k in range(n_frames):
        for pl in range(len(want[k])):
            bad = np.argwhere(want[k][pl] != got[k][pl])
            assert not len(bad), "frame %d plane %d differs at %s (%d pixels)" % (k, pl, bad[0], len(bad))
    st = hk.run.last_device_stats
    assert len(st) == 2 and st[0][0] == (n_frames + 1) // 2 and st[1][0] == n_frames // 2, st       # frames in turn
    # every inter frame predicts from the three frames before it: at least one of them ended on the other device
    assert st[0][1] + st[1][1] >= n_frames - 1, st


def test_rows_cross_to_the_other_device_while_their_frame_is_still_ending(ctx):
    """Dav1dHipGlueOptions.row_progress with two devices: the rows a frame publishes while its last stage runs (dav1d_hip_frame_set_progress_callback
    -> progress[1], reference src/thread_task.c:888-896) also cross, band by band, to the device on which a queued frame predicts from the picture
    (dav1d_hip_picture_copy_peer_rows) — dav1d's frame threads start on published rows (src/thread_task.c:416-433); across devices what starts
    early is the transfer.  The consumer finds the picture nearly there, sends the remainder and makes the mirror's tiled twin itself.  A chain
    with restoration (the banded last stage), every picture equal to dav1d's, bands counted."""
    if n_devices_here(ctx) < 2:
        pytest.skip("one device here")
    w, h, bpc = 320, 1100, 10                 # (tall enough for the last stage to run in bands)
    n_frames = 4 if ctx.backend == "emu" else 9
    kw = dict(tiles=(1, 1))
    _, _, want = hk.run(hk.params(w, h, bpc, n_frames, mode=0, **kw), hip_lib_path(ctx))
    _, _, got = hk.run(hk.params(w, h, bpc, n_frames, mode=1, n_devices=2, row_progress=1, **kw), hip_lib_path(ctx))
    for k in range(n_frames):
        for pl in range(len(want[k])):
            bad = np.argwhere(want[k][pl] != got[k][pl])
            assert not len(bad), "frame %d plane %d differs at %s (%d pixels)" % (k, pl, bad[0], len(bad))
    st, bands = hk.run.last_device_stats, hk.run.last_band_copies
    assert len(st) == 2 and st[0][1] + st[1][1] >= n_frames - 1, st
    assert hk.run.last_row_publications > n_frames, hk.run.last_row_publications       # rows came band by band
    assert sum(bands) > 0, (bands, st)            # ... and crossed that way at least once
