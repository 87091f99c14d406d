"""dav1d's REAL pass 1 in front of the backend (VERDICT round 3, item 1 / SURVEY 8 row "task lists produced by dav1d's own CPU
msac / decode.c").

Nothing is injected and nothing comes from a generator of the product: tests/av1_obu.py writes AV1 bitstreams — sequence header, frame
headers with every coding tool switched on at random (switchable interpolation, all compound types, OBMC, local + global warps,
inter-intra, palette + intra block copies, filter-intra, CfL, segmentation with features, delta_q / delta_lf, lossless segments,
quantiser matrices, both transform-set sizes, switchable transform size, CDEF with up to 8 strengths, switchable restoration,
super-resolution, reference scaling, film grain, hidden frames + show_existing_frame, intra-only frames, non-uniform tiles) and tile
payloads of seeded random bytes — and the reference's own front end decodes them: dav1d_send_data / dav1d_parse_obus /
dav1d_submit_frame, dav1d_msac_*, decode_b, decode_coefs, read_restoration_info, dav1d_create_lf_mask_*, refmvs, CDF adaptation, on
dav1d's worker threads under dav1d's task loop (oracle/_ref_hooked).  An arithmetic decoder decodes any byte string into valid
syntax (the reference's own fuzzer relies on it, tests/libfuzzer/dav1d_fuzzer.c), so pass 1's output has the statistics of the
default CDFs: Av1Block / cbi / cf / palettes / Av1Filter masks / cdef_idx / restoration units as real streams produce them.

Each stream is decoded twice: by dav1d alone (mode 0: no hook but an error recorder; pass 2, filters and film grain are the
reference's C code), and with the glue of INTEGRATION.md behind dav1d's pass 1 (mode 1: Dav1dPicAllocator on
dav1d_hip_host_picture_*, dav1d_hip_lister_tile_sbrow / _filter_sbrow from the tile and filter tasks, dav1d_hip_frame_end, film
grain on the output picture by dav1d_hip_fg_apply).  Every picture dav1d_get_picture hands out must be identical.  A payload dav1d
rejects (4:2:2 forbids some partitions; an intra block copy can point nowhere) is re-rolled from the failing byte on until dav1d
accepts the stream (stream_util.repair) — both modes then see the same, valid stream."""
import collections
import os

import numpy as np
import pytest

import util
import av1_obu
import stream_util as su
import hooked_util as hk
from dav1d_amd import _lib

pytestmark = pytest.mark.skipif(su.lib() is None, reason="needs oracle/_ref_hooked (the reference build with the hook patch)")

LAYOUTS = {0: "400", 1: "420", 2: "422", 3: "444"}
SIZES = [(256, 192), (200, 136), (320, 176), (130, 98), (448, 264), (384, 121), (264, 250)]
TOTAL = collections.Counter()
SEEN = collections.Counter()


def hip_lib_path(ctx):
    return util.emu_lib_path() if ctx.backend == "emu" else _lib.DEFAULT_PATH


def config_of(seed):
    """the sweep: layout x bit depth x superblock size x frame size, tiles drawn per frame (1 - 4 columns / rows)"""
    layout = (1, 3, 0, 2)[seed % 4]
    bpc = (8, 10, 12)[seed % 3]
    w, h = SIZES[(seed // 4) % len(SIZES)]
    return layout, bpc, w, h, bool(seed & 8)


class Unrepairable(Exception):
    pass


def run_seed(ctx, seed, n_frames=7, knobs=None, **dec):
    layout, bpc, w, h, sb128 = config_of(seed)
    sw = av1_obu.make_stream(w, h, layout, bpc, n_frames, seed, sb128=sb128, knobs=knobs)
    lib = hip_lib_path(ctx)
    try:
        su.repair(sw, lib, max_rounds=8000)
    except AssertionError as e:
        # the WRITER's limit, not the backend's (dav1d alone decodes in repair): a payload whose re-rolls keep running into syntax dav1d
        # rejects (seed 1763, 4:2:2 12-bit: one in 2,048).  The sweeps count such streams and go on; everything else fails as before
        if "still rejected" in str(e):
            raise Unrepairable("seed %d: %s" % (seed, e))
        raise
    units = [u["data"] for u in sw.units]
    want = su.decode(units, 0, lib)
    assert not want["errors"] and len(want["pictures"]) >= n_frames - 2
    got = su.decode(units, 1, lib, free_listing=seed & 1, pack=not seed & 16, row_progress=(seed >> 5) & 1, **dec)
    diff = su.compare(want, got)
    assert diff is None, "seed %d (%s %d-bit %dx%d sb%d): %s" % (seed, LAYOUTS[layout], bpc, w, h, 128 if sb128 else 64, diff)
    TOTAL.update(got["hist"])
    SEEN[(LAYOUTS[layout], bpc, 128 if sb128 else 64)] += 1
    for u in sw.units:
        if u["frame"] is not None:
            SEEN["tiles %dx%d" % (u["frame"].tile_cols, u["frame"].tile_rows)] += 1
    return got


def report(title):
    print("\n%s: streams by (layout, bpc, superblock) and frames by tiling: %s" % (title, dict(sorted((str(k), v) for k, v in SEEN.items()))))
    print("tools pass 1 of dav1d really produced (blocks / frames counted by the reference's own block walk):")
    for k in su.HIST:
        print("    %-24s %d" % (k, TOTAL[k]))


MUST = ["b_intra", "b_inter", "b_skip", "b_cfl", "b_filter_intra", "b_directional", "b_smooth", "b_paeth", "b_comp_avg", "b_comp_wavg",
        "b_comp_seg", "b_comp_wedge", "b_interintra", "b_obmc", "b_local_warp", "b_globalmv", "b_dual_filter", "b_tx_split", "b_tx64",
        "b_sub8x8_chroma", "tx_non_dct", "lr_wiener", "lr_sgr", "cdef_nonzero_idx", "frames_super_res", "frames_scaled_refs",
        "frames_film_grain", "frames_segmented", "frames_delta_lf", "b_palette_y", "b_intrabc"]


def assert_covered():
    missing = [k for k in MUST if not TOTAL[k]]
    assert not missing, "never produced by the streams: %s" % missing


EMU_SEEDS = list(range(1, 13))


@pytest.mark.parametrize("seed", EMU_SEEDS)
def test_random_streams_through_dav1ds_own_front_end(ctx, seed):
    if ctx.backend != "emu":
        pytest.skip("the GPU run takes the sweep below")
    run_seed(ctx, seed, n_frames=6)


@pytest.mark.parametrize("seed", EMU_SEEDS)
def test_random_streams_over_two_devices_in_one_process(ctx, seed):
    """the same twelve streams with the binding's frames ending on two devices in turn (Dav1dHipGlueOptions.n_devices = 2): super-resolution
    (two pictures per frame), scaled references, film grain on the output's device, show-existing frames, and frames dav1d drops"""
    import ctypes as C
    if C.CDLL(hip_lib_path(ctx)).dav1d_hip_device_count() < 2:
        pytest.skip("one device here")
    got = run_seed(ctx, seed, n_frames=6, n_devices=2)
    st = got["device_stats"]
    assert len(st) == 2 and st[0][0] and st[1][0], st


def test_screen_content_stream_palette_and_intra_block_copy(ctx):
    """key / intra-only frames with palettes and intra block copies (allow_screen_content_tools on every frame)"""
    k = av1_obu.Knobs(screen_content=1.0, intrabc=1.0, intra_only=0.5, super_res=0.0)
    got = run_seed(ctx, 4001, n_frames=5, knobs=k)
    assert got["hist"]["b_palette_y"] > 0 and got["hist"]["b_intrabc"] > 0, got["hist"]


def test_wavefront_steps_come_from_cells_their_tile_has_cleared(ctx, monkeypatch):
    """The lister's cell maps are recycled from frame to frame uncleared; a tile clears its share when its first superblock row is listed.
    decode_b lets the source rectangle of an intra block copy end at the tile's last row / column (src/decode.c:1290-1336): the cell
    behind it belongs to the next tile — listed by another thread, or still holding whatever the map held.  DAV1D_HIP_LISTER_POISON fills
    the maps with 0xffff at the start of a frame, so a step drawn from such a cell is refused (-ERANGE) instead of differing from run to
    run: that is how one stream in 3,000 of the GPU sweeps failed (round 6).  Streams 4001, 4002 and 4005 failed here before the fix."""
    monkeypatch.setenv("DAV1D_HIP_LISTER_POISON", "1")
    k = av1_obu.Knobs(screen_content=1.0, intrabc=1.0, intra_only=0.5, super_res=0.0)
    for seed in (4001, 4002, 4005):
        got = run_seed(ctx, seed, n_frames=5, knobs=k)
        assert got["hist"]["b_intrabc"] > 0
    for seed in (2, 7):
        run_seed(ctx, seed, n_frames=5)


def test_filters_over_a_full_copy_give_the_same_pictures(ctx, monkeypatch):
    """CDEF and restoration normally bring over only the units they do not list (cdef.hip cdef_fill_unlisted_kernel, frame.hip
    copy_unrestored_planes); the A/B switch puts the whole-picture copy back underneath: both are dav1d's pictures"""
    monkeypatch.setenv("DAV1D_HIP_FILTER_FULL_COPY", "1")
    for seed in (3, 11):
        run_seed(ctx, seed, n_frames=5)


def test_cdef_from_unit_records_gives_the_same_pictures(ctx, monkeypatch):
    """by default the filter lister hands CDEF over as one record per unit row of a 64-pixel column and the device makes the unit records
    (cdef.hip cdef_expand_kernel); DAV1D_HIP_CDEF_ROWS=0 keeps the host-made unit records: both are dav1d's pictures"""
    monkeypatch.setenv("DAV1D_HIP_CDEF_ROWS", "0")
    for seed in (5, 12):
        run_seed(ctx, seed, n_frames=5)


N_SWEEP = int(os.environ.get("DAV1D_STREAM_SEEDS", "208"))
SWEEP_PARTS = [list(range(lo, min(lo + 16, 100 + N_SWEEP))) for lo in range(100, 100 + N_SWEEP, 16)]


@pytest.mark.gpu
@pytest.mark.parametrize("seeds", SWEEP_PARTS, ids=["seeds-%d-%d" % (p[0], p[-1]) for p in SWEEP_PARTS])
def test_sweep_of_streams_on_the_gpu(seeds):
    """>= 200 seeds over 8 / 10 / 12 bit x 4:0:0 / 4:2:0 / 4:2:2 / 4:4:4 x 64- / 128-pixel superblocks x 1 - 4 tile columns / rows
    (drawn per frame), on the MI355X, in cases of 16 seeds (a failure names its case; the assertion names the seed).
    DAV1D_STREAM_SEEDS overrides the count."""
    ctx = util.make_context("hip")
    ctx.backend = "hip"
    try:
        for seed in seeds:
            try:
                run_seed(ctx, seed, n_frames=7)
            except Unrepairable as e:
                SEEN["unrepairable"] += 1
                print(e)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_sweep_of_streams_on_the_gpu_covered_every_tool():
    """runs behind the cases above (same process: the histogram of the tools pass 1 of dav1d really produced has accumulated)"""
    if sum(v for k, v in SEEN.items() if isinstance(k, tuple)) < 32:
        pytest.skip("the sweep did not run in this process (or with too few seeds)")
    report("GPU sweep of %d streams" % N_SWEEP)
    assert_covered()
    assert SEEN["unrepairable"] * 200 <= N_SWEEP, "%d streams of %d the writer could not repair" % (SEEN["unrepairable"], N_SWEEP)
    layouts = {k[0] for k in SEEN if isinstance(k, tuple)}
    assert layouts == {"400", "420", "422", "444"}
    assert {k[1] for k in SEEN if isinstance(k, tuple)} == {8, 10, 12} and {k[2] for k in SEEN if isinstance(k, tuple)} == {64, 128}


@pytest.mark.gpu
def test_larger_streams_on_the_gpu():
    """1080p / 4K-class frames: many superblocks per tile, 128-pixel superblocks, up to 4 x 4 tiles, worker threads as in production"""
    ctx = util.make_context("hip")
    ctx.backend = "hip"
    try:
        for seed, (w, h, layout, bpc) in enumerate([(1920, 1080, 1, 8), (2048, 1152, 1, 10), (1280, 720, 3, 12), (3840, 2160, 1, 10)]):
            k = av1_obu.Knobs(bytes_per_pixel=0.6)
            sw = av1_obu.make_stream(w, h, layout, bpc, 5, 9000 + seed, sb128=bool(seed & 1), knobs=k)
            lib = hip_lib_path(ctx)
            su.repair(sw, lib, threads=16, max_rounds=200)
            units = [u["data"] for u in sw.units]
            want = su.decode(units, 0, lib, threads=16, frame_delay=4)
            got = su.decode(units, 1, lib, threads=16, frame_delay=4)
            diff = su.compare(want, got)
            assert diff is None, "%dx%d: %s" % (w, h, diff)
            TOTAL.update(got["hist"])
    finally:
        ctx.close()


def test_histogram_covers_the_tools(ctx):
    """(runs after the sweeps of this module) every tool the lister restates has been met in pass 1's real output"""
    if ctx.backend != "emu" or sum(v for k, v in SEEN.items() if isinstance(k, tuple)) < 10:
        pytest.skip("the sweep of this backend did not run (the GPU sweep checks its own histogram)")
    report("streams so far")
    assert_covered()
