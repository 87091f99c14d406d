"""End-to-end leg of the benchmark: hand-off arrays -> pass-2 lister (host threads) -> device.  BENCH / TEST INFRASTRUCTURE (it drives the
generator of synthetic pass-1 output, tests/synth): lives next to the tests, not in the product package.

What the headline `value` of bench.py leaves out by definition (lists resident in HBM) is measured here: one synthetic 8K
frame's pass-1 output (Av1Block / cbi / cf, dav1d_synth_frame) is listed by dav1d_hip_lister_tile_sbrow() from a pool of
host threads (one per tile, as dav1d's pass-2 workers would), every submit prepares its chunk of the device lists on the
submitting thread, the coefficient arena crosses the host link, and dav1d_hip_frame_end() launches the frame.

    python tests/e2e.py [--frames N] [--threads T] [--tile-cols C]
"""
import argparse
import ctypes as C
import json
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dav1d_amd import _lib, api  # noqa: E402

# the generator of synthetic pass-1 output is test infrastructure with a library of its own (tests/synth/libdav1d_synth.so)
import synth_lib  # noqa: E402

SIZE_MUL = [(4, 4), (6, 5), (8, 6), (12, 8)]      # ss_size_mul, reference src/decode.c:2416-2421


def jnt_weights(cur_poc, ref_poc, order_hint_bits):
    """f->jnt_weights as dav1d_decode_frame_init() derives them from the order hints (reference src/decode.c:3083-3120):
    the distance-weighted compound weights of AV1 (spec 7.11.3.15)."""
    def diff(a, b):
        if not order_hint_bits:
            return 0
        m = 1 << (order_hint_bits - 1)
        d = a - b
        return (d & (m - 1)) - (d & m)
    wt = ((2, 3), (2, 5), (2, 7))
    look = ((9, 7), (11, 5), (12, 4), (13, 3))
    out = [[0] * 7 for _ in range(7)]
    for i in range(7):
        for j in range(i + 1, 7):
            d1 = min(abs(diff(ref_poc[i], cur_poc)), 31)
            d0 = min(abs(diff(ref_poc[j], cur_poc)), 31)
            order = int(d0 <= d1)
            k = 0
            while k < 3:
                c0, c1 = wt[k][order], wt[k][1 - order]
                if (d0 > d1 and d0 * c0 < d1 * c1) or (d0 <= d1 and d0 * c0 > d1 * c1):
                    break
                k += 1
            out[i][j] = look[k][order]
    return out


CUR_POC, REF_POC, ORDER_HINT_BITS = 8, (7, 6, 4, 2, 9, 10, 12), 5       # the frame order the synthetic frames pretend to have


class HandOff:
    """The per-frame arrays of dav1d's frame threading, sized as dav1d_decode_frame_init() sizes them
    (reference src/decode.c:2839-2895, 3002-3015), as numpy buffers."""

    def __init__(self, w, h, layout, bpc, sb128=True, tile_cols=1, tile_rows=1):
        self.w, self.h, self.layout, self.bpc, self.sb128 = w, h, layout, bpc, sb128
        hbd = bpc > 8
        self.bw, self.bh = ((w + 7) >> 3) << 1, ((h + 7) >> 3) << 1
        self.sb128w, self.sb128h = (self.bw + 31) >> 5, (self.bh + 31) >> 5
        n = self.sb128w * self.sb128h
        sm = SIZE_MUL[layout]
        self.b = np.zeros(n * 32 * 32 * 32, np.uint8)
        self.cbi = np.zeros(n * sm[0] * 32 * 32 // 4, np.int16)
        self.cf = np.zeros(((n * sm[0]) << hbd) * 128 * 128 // 2, np.uint8)
        sb = 128 if sb128 else 64
        sbw, sbh = (w + sb - 1) // sb, (h + sb - 1) // sb
        self.sbh = sbh

        def uniform(n_sb, k):
            k = max(1, min(k, n_sb))
            size = (n_sb + k - 1) // k
            return list(range(0, n_sb, size)) + [n_sb]
        self.cols, self.rows = uniform(sbw, tile_cols), uniform(sbh, tile_rows)
        # f->frame_thread.tile_start_off, src/decode.c:2803-2813
        sb_step4 = (32 if sb128 else 16) * 4
        tso = []
        for tr in range(len(self.rows) - 1):
            row_off = self.rows[tr] * sb_step4 * self.sb128w * 128
            b_diff = (self.rows[tr + 1] - self.rows[tr]) * sb_step4
            for tc in range(len(self.cols) - 1):
                tso.append(row_off + b_diff * self.cols[tc] * sb_step4)
        self.tile_start_off = np.array(tso, np.uint32)
        d = _lib.FrameDesc()
        d.w, d.h, d.layout, d.bpc, d.sb128, d.intra_edge_filter, d.is_inter = w, h, layout, bpc, int(sb128), 1, 1
        d.n_tile_cols, d.n_tile_rows = len(self.cols) - 1, len(self.rows) - 1
        for i, v in enumerate(self.cols):
            d.col_start_sb[i] = v
        for i, v in enumerate(self.rows):
            d.row_start_sb[i] = v
        d.b4_stride = (self.bw + 31) & ~31
        d.b, d.cbi, d.tile_start_off = self.b.ctypes.data, self.cbi.ctypes.data, self.tile_start_off.ctypes.data
        d.cf_align64 = 1
        jw = jnt_weights(CUR_POC, REF_POC, ORDER_HINT_BITS)
        for i in range(7):
            d.ref_w[i], d.ref_h[i] = w, h
            for j in range(7):
                d.jnt_weights[i][j] = jw[i][j]
        self.desc = d


def c2_params(seed, n_refs=3):
    """SURVEY 8d config C2's itx+mc subset as generator settings: block mix 64 / 32 / 16 / 8 / 4 = 20 / 30 / 30 / 15 / 5 % by area
    (split probabilities per level: 1 - 0.2, 1 - 0.3 / 0.8, ...), 25 % compound average, every block coded, no intra blocks."""
    sp = synth_lib.SynthParams()
    sp.seed = seed
    sp.intra_pct, sp.skip_pct = 0, 0
    sp.compound_pct, sp.masked_compound = 25, 0
    sp.tx_split_pct, sp.alt_txtp_pct, sp.eob_none_pct = 0, 30, 0
    sp.mv_range, sp.far_mv_pct, sp.n_refs = 512, 5, n_refs
    for i, v in enumerate((100, 80, 63, 50, 25)):
        sp.split_pct[i] = v
    sp.rect_pct = 0
    sp.fixed_bl = -1
    sp.cf_align64 = 1
    return sp


def run(ctx, w=7680, h=4320, bpc=10, frames=6, threads=None, tile_cols=4, seed=0xE2E, check=None, intra_pct=0, key_frame=False, tile_rows=1,
        native_threads=True, packed=False, intrabc_pct=0):
    """Returns the measurement dict.  check: optional callable(handoff, desc, planes) -> str used as the parity gate.
    packed: the packing lister (Dav1dHipFrameDesc.cf: the eob + 1 values per block go to the frame's own arena, the host arena is
    left zeroed) instead of the dense arena crossing the host link; every frame gets a fresh copy of pass 1's coefficients."""
    layout = api.LAYOUT_I420
    ho = HandOff(w, h, layout, bpc, True, tile_cols, tile_rows)
    sp = c2_params(seed)
    sp.intra_pct = 100 if key_frame else intra_pct
    sp.intrabc_pct = intrabc_pct          # key frames: that share of the blocks are intra block copies (screen content)
    if key_frame:
        ho.desc.is_inter = 0
    t0 = time.perf_counter()
    rc = synth_lib.synth_frame(ho.desc, sp, ho.cf.ctypes.data, ho.cf.nbytes, len(ho.cbi), None, 0)
    assert rc == 0, rc
    t_synth = time.perf_counter() - t0
    rng = np.random.default_rng(seed)
    refs = []
    for _ in range(3):
        r = ctx.picture(w, h, layout, bpc)
        for pl in range(3):
            rows, cols = r.padded_shape(pl)
            r.upload(pl, rng.integers(0, 1 << bpc, size=(rows, cols), dtype=np.uint16).astype(r.dtype))
        refs.append(r)
    refs7 = [refs[i % 3] for i in range(7)]
    cur = ctx.picture(w, h, layout, bpc)
    n_tcols, n_trows = ho.desc.n_tile_cols, ho.desc.n_tile_rows
    n_tiles = n_tcols * n_trows
    threads = threads or n_tiles
    coef = ctx.buffer(ho.cf.nbytes if not packed else 4096)
    import torch
    pinned = torch.from_numpy(ho.cf).pin_memory() if torch.cuda.is_available() and not packed else None
    cf_copies = [np.array(ho.cf, copy=True) for _ in range(frames)] if packed else None
    packed_bytes = 0
    nb = C.c_size_t()
    blob = ctx.lib.dav1d_hip_lister_const_masks(C.byref(nb))
    prep = mask = None
    res = {"list_ms": [], "h2d_ms": [], "frame_end_ms": [], "total_ms": []}
    steps = 0
    planes = None
    with ThreadPoolExecutor(threads) as ex, ThreadPoolExecutor(1) as ex2:
        for it in range(frames):
            t_a = time.perf_counter()
            frame = ctx.frame(cur, refs7)
            lh = C.c_void_p()
            desc = ho.desc
            if packed:
                desc = _lib.FrameDesc.from_buffer_copy(ho.desc)
                desc.cf = cf_copies[it].ctypes.data
            assert ctx.lib.dav1d_hip_lister_create(C.byref(lh), C.byref(desc), frame.h) == 0

            def tile(k):
                tr, tc = divmod(k, n_tcols)
                for sby in range(ho.rows[tr], ho.rows[tr + 1]):
                    rc2 = ctx.lib.dav1d_hip_lister_tile_sbrow(lh, tr, tc, sby)
                    assert rc2 == 0, rc2
            # the coefficient arena crosses the host link every frame (the kernels consume = zero it), while the listing runs
            def h2d():
                if packed:
                    return 0.0
                t = time.perf_counter()
                src = pinned.data_ptr() if pinned is not None else ho.cf.ctypes.data
                assert ctx.lib.dav1d_hip_upload(ctx.h, coef.ptr, src, ho.cf.nbytes) == 0
                return (time.perf_counter() - t) * 1e3
            up = ex2.submit(h2d)
            if native_threads:
                # the library's own threads walk the tiles (dav1d_hip_lister_run): no interpreter lock between the tile-sbrows
                rc2 = ctx.lib.dav1d_hip_lister_run(lh, threads)
                assert rc2 == 0, rc2
            else:
                list(ex.map(tile, range(n_tiles)))
            t_b = time.perf_counter()
            h2d_ms = up.result()
            if prep is None:
                prep = ctx.buffer(ctx.lib.dav1d_hip_lister_prep_elems(lh) * 2 + 4096)
                mask = ctx.buffer(ctx.lib.dav1d_hip_lister_mask_bytes(lh) + 4096)
                mask.upload(np.ctypeslib.as_array((C.c_uint8 * nb.value).from_address(blob)))
            t_c = time.perf_counter()
            if packed:
                packed_bytes = int(ctx.lib.dav1d_hip_frame_coef_bytes(frame.h))
            frame.end(None if packed else coef, prep, mask)
            t_d = time.perf_counter()
            steps = ctx.lib.dav1d_hip_lister_steps(lh)
            if it == frames - 1 and check is not None:
                planes = [cur.download(pl) for pl in range(3)]
            ctx.lib.dav1d_hip_lister_destroy(lh)
            frame.destroy()
            if it:          # the first frame pays the pools' allocations
                res["list_ms"].append((t_b - t_a) * 1e3)
                res["h2d_ms"].append(h2d_ms)
                res["frame_end_ms"].append((t_d - t_c) * 1e3)
                res["total_ms"].append((t_d - t_a) * 1e3)
    out = {k: round(float(np.median(v)), 3) for k, v in res.items() if v}
    out["frames"] = frames - 1
    out["host_threads"] = threads
    out["tiles"] = n_tiles
    out["wavefront_steps"] = int(steps)
    out["coef_bytes_per_frame"] = int(packed_bytes if packed else ho.cf.nbytes)
    out["value"] = round(w * h / (out["total_ms"] * 1e-3) / 1e6, 1) if "total_ms" in out else None
    out["unit"] = "Mpixels/s"
    out["synth_seconds"] = round(t_synth, 2)
    kind = "key frame (every block intra)" if key_frame else "inter frame" if not intra_pct else "inter frame, %d %%%% intra blocks" % intra_pct
    out["workload"] = ("%dx%d 4:2:0 %d-bit " + kind + " from pass-1 hand-off arrays: lister on %d host threads" + (" of the library" if native_threads else "") + " over %d x %d tiles, chunk "
                       "preparation + upload on the submitting threads, " + ("the packing lister's values (eob + 1 per block) in the frame's own arena, " if packed else "dense coefficient arena over the host link meanwhile (h2d_ms), ") +
                       "frame_end = gather + the frame's launches + sync") % (w, h, bpc, threads, n_tcols, n_trows)
    if check is not None and planes is not None:
        out["parity"] = check(ho, planes, refs)
    for o in refs + [cur, coef] + ([prep, mask] if prep is not None else []):
        o.free()
    return out


def cpu_quota():
    """(cores the container may use per scheduling period or None, throttled periods so far or None): cgroup v2 cpu.max / cpu.stat"""
    cores = thr = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        cores = None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                thr = int(line.split()[1])
    except (OSError, ValueError):
        pass
    return cores, thr


def host_threads(n_units, asked=None):
    """Listing threads for the end-to-end legs: what was asked for, else as many as the host lets run at once — the CPU quota of the
    container when there is one (more threads than that are throttled as a group: 16 cores' worth per 100 ms on the MI355X boxes of
    this pool, after which every thread of the process stops until the period is over), else the CPUs this process may run on, 64 at most."""
    if asked:
        return min(asked, n_units)
    cores, _ = cpu_quota()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(64, avail)
    if cores:
        n = max(1, min(n, int(cores * 1.5)))        # a little over the quota: threads also wait (locks, the queue's tail)
    return max(1, min(n, n_units))


def run_pipelined(ctx, desc, cf_frames, w, h, layout, bpc, refs, threads, depth=2, warm=2, filters=None):
    """Frames in flight, as dav1d's frame threading has them: frame n + 1 is listed (host threads) while frame n is on the device.

    cf_frames: one HOST coefficient arena per frame (f->frame_thread.cf of that frame's context; the packing lister consumes it:
    Dav1dHipFrameDesc.cf).  `depth` frames own a picture, a prep and a mask arena each.  filters: None, or dict(fd=FilterDesc,
    lvl_host=level cache, b4_stride, lut_e, lut_i, damping) — the frame then runs its in-loop filters too (the level cache goes
    up per frame on a context of its own, so that the copy does not queue behind the frame on the device).
    The listing thread (this one) and the ending thread hand frames over a queue; a picture slot is reused when its frame is
    through.  Returns (stats, planes of the last frame): ms_per_frame is the sustained rate between the ends of frame warm - 1
    and the last one; list / end times are per-frame medians of the two threads' own clocks."""
    import queue
    import threading
    lib = ctx.lib
    frames = len(cf_frames)
    up_ctx = api.Context(ctx.device, lib_path=ctx.lib_path) if filters else None
    lvl_keep = None
    if filters:
        # the level cache leaves from page-locked memory, as a picture allocator callback of the integration would provide it
        try:
            import torch
            if torch.cuda.is_available():
                lvl_keep = torch.from_numpy(np.ascontiguousarray(filters["lvl_host"])).pin_memory()
                filters = dict(filters, lvl_host=lvl_keep.numpy())
        except ImportError:
            pass
    slots = []
    for _ in range(depth):
        slots.append(dict(cur=ctx.picture(w, h, layout, bpc), prep=None, mask=None,
                          lvl=ctx.buffer(len(filters["lvl_host"])) if filters else None))
    free_slots = queue.Queue()
    for sl in slots:
        free_slots.put(sl)
    todo = queue.Queue()
    nb = C.c_size_t()
    blob = lib.dav1d_hip_lister_const_masks(C.byref(nb))
    const_masks = np.ctypeslib.as_array((C.c_uint8 * nb.value).from_address(blob))
    t_end, end_ms, err, destroy_ms, begin_ms, t_listed = [], [], [], [], [], []
    last = {}

    def ender():
        while True:
            job = todo.get()
            if job is None:
                return
            frame, lh, sl, it = job
            try:
                t = time.perf_counter()
                filtered = frame.end(None, sl["prep"], sl["mask"])
                now = time.perf_counter()
                t_end.append(now)
                end_ms.append((now - t) * 1e3)
                if it == frames - 1:
                    pic = api.DevicePicture.view(ctx, filtered, w, h, layout, bpc) if filters else sl["cur"]
                    last["planes"] = [pic.download(pl) for pl in range(1 if layout == 0 else 3)]
            except Exception as e:          # noqa: BLE001 - reported by the caller
                err.append(e)
            t = time.perf_counter()
            lib.dav1d_hip_lister_destroy(lh)
            frame.destroy()
            destroy_ms.append((time.perf_counter() - t) * 1e3)
            free_slots.put(sl)
    th = threading.Thread(target=ender)
    th.start()
    list_ms, flist_ms, wait_ms = [], [], []
    d = _lib.FrameDesc.from_buffer_copy(desc)
    cpu0 = thr0 = None
    try:
        for it in range(frames):
            if it == warm:
                cpu0, (_, thr0) = time.process_time(), cpu_quota()
            t0 = time.perf_counter()
            sl = free_slots.get()
            t_a = time.perf_counter()
            frame = ctx.frame(sl["cur"], refs)
            d.cf = cf_frames[it].ctypes.data
            lh = C.c_void_p()
            assert lib.dav1d_hip_lister_create(C.byref(lh), C.byref(d), frame.h) == 0
            begin_ms.append((time.perf_counter() - t_a) * 1e3)
            if filters:
                assert lib.dav1d_hip_upload(up_ctx.h, sl["lvl"].ptr, filters["lvl_host"].ctypes.data, len(filters["lvl_host"])) == 0
            rc = lib.dav1d_hip_lister_run_frame(lh, C.byref(filters["fd"]), threads) if filters else lib.dav1d_hip_lister_run(lh, threads)
            assert rc == 0, rc
            t_b = time.perf_counter()
            if filters:
                frame.set_filters(sl["lvl"], filters["b4_stride"], filters["lut_e"], filters["lut_i"], filters["damping"])
            t_c = time.perf_counter()
            coef_bytes = int(lib.dav1d_hip_frame_coef_bytes(frame.h))
            need_prep, need_mask = lib.dav1d_hip_lister_prep_elems(lh) * 2 + 4096, lib.dav1d_hip_lister_mask_bytes(lh) + 4096
            if sl["prep"] is None or sl["prep"].nbytes < need_prep:
                sl["prep"] = ctx.buffer(need_prep + need_prep // 4)
            if sl["mask"] is None or sl["mask"].nbytes < need_mask:
                sl["mask"] = ctx.buffer(need_mask + need_mask // 4)
                assert lib.dav1d_hip_upload((up_ctx or ctx).h, sl["mask"].ptr, const_masks.ctypes.data, nb.value) == 0
            todo.put((frame, lh, sl, it))
            t_listed.append(time.perf_counter())
            if it >= warm:
                wait_ms.append((t_a - t0) * 1e3)
                list_ms.append((t_b - t_a) * 1e3)
                flist_ms.append((t_c - t_b) * 1e3)
    finally:
        todo.put(None)
        th.join()
    if err:
        raise err[0]
    n = frames - warm
    cpu1, (quota, thr1) = time.process_time(), cpu_quota()
    gaps = np.diff(np.array(t_end[warm - 1:])) * 1e3
    out = {"ms_per_frame": round((t_end[-1] - t_end[warm - 1]) / n * 1e3, 3), "frames": n, "frames_in_flight": depth,
           "ms_between_frame_ends": {"median": round(float(np.median(gaps)), 3), "max": round(float(gaps.max()), 3)},
           "list_ms": round(float(np.median(list_ms)), 3), "frame_end_ms": round(float(np.median(end_ms[warm:])), 3),
           "slot_wait_ms": round(float(np.median(wait_ms)), 3), "host_threads": threads,
           "packed_coef_bytes_per_frame": coef_bytes,
           # what the host side costs: CPU time of the whole process (listing threads, the ending thread, this harness) per frame.  Where
           # the container has a CPU quota, sustained ms_per_frame cannot go below this divided by the quota, however many threads run.
           "host_cpu_ms_per_frame": round((cpu1 - cpu0) / n * 1e3, 1) if cpu0 is not None else None,
           "host_cpu_quota_cores": quota, "throttled_periods_during_run": (thr1 - thr0) if thr0 is not None and thr1 is not None else None}
    if os.environ.get("DAV1D_HIP_E2E_TRACE"):
        r = lambda v: [round(float(x), 2) for x in v]
        out["trace"] = {"list": r(list_ms), "end": r(end_ms), "destroy": r(destroy_ms), "begin": r(begin_ms),
                        "listed_at": r((np.array(t_listed) - t_listed[0]) * 1e3), "ended_at": r((np.array(t_end) - t_listed[0]) * 1e3)}
    out["value"] = round(w * h / (out["ms_per_frame"] * 1e-3) / 1e6, 1)
    out["unit"] = "Mpixels/s"
    for sl in slots:
        for k in ("cur", "prep", "mask", "lvl"):
            if sl[k] is not None:
                sl[k].free()
    if up_ctx:
        up_ctx.close()
    return out, last.get("planes")


def run_sustained(ctx, w=7680, h=4320, bpc=10, frames=10, threads=None, tile_cols=4, tile_rows=1, seed=0xE2E, check=None, depth=2, warm=2,
                  key_frame=False):
    """The reconstruction route with frames in flight (run_pipelined): the synthetic inter frame of run() (key_frame: its key frame,
    every block intra), every frame with a coefficient arena of its own that the packing lister consumes; nothing dense crosses the
    host link."""
    layout = api.LAYOUT_I420
    ho = HandOff(w, h, layout, bpc, True, tile_cols, tile_rows)
    sp = c2_params(seed)
    if key_frame:
        sp.intra_pct = 100
        ho.desc.is_inter = 0
    rc = synth_lib.synth_frame(ho.desc, sp, ho.cf.ctypes.data, ho.cf.nbytes, len(ho.cbi), None, 0)
    assert rc == 0, rc
    rng = np.random.default_rng(seed)
    refs = []
    for _ in range(3):
        r = ctx.picture(w, h, layout, bpc)
        for pl in range(3):
            rows, cols = r.padded_shape(pl)
            r.upload(pl, rng.integers(0, 1 << bpc, size=(rows, cols), dtype=np.uint16).astype(r.dtype))
        refs.append(r)
    n_tiles = ho.desc.n_tile_cols * ho.desc.n_tile_rows
    threads = host_threads(n_tiles, threads)
    cfs = [ho.cf.copy() for _ in range(frames)]
    out, planes = run_pipelined(ctx, ho.desc, cfs, w, h, layout, bpc, [refs[i % 3] for i in range(7)], threads, depth, warm)
    out["tiles"] = n_tiles
    out["host_arena_left_zero"] = not any(bool(c.any()) for c in cfs)
    out["workload"] = ("%dx%d 4:2:0 %d-bit %s frame from pass-1 hand-off arrays, %d frames in flight: the packing lister on %d library "
                       "threads over %d x %d tiles (eob + 1 values per block into the frame's pinned arena, host arena zeroed), chunk "
                       "preparation on the listing threads, frame_end (transfer + gather + launches + sync) of frame n under the "
                       "listing of frame n + 1" % (w, h, bpc, "key" if key_frame else "inter", depth, threads, ho.desc.n_tile_cols, ho.desc.n_tile_rows))
    if check is not None and planes is not None:
        out["parity"] = check(ho, planes, refs)
    for o in refs:
        o.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--tile-cols", type=int, default=4)
    ap.add_argument("--tile-rows", type=int, default=1)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--intra-pct", type=int, default=0)
    ap.add_argument("--key-frame", type=int, default=0)
    ap.add_argument("--sustained", type=int, default=0, help="frames in flight (0: one frame at a time, dense coefficients)")
    a = ap.parse_args()
    ctx = api.Context(0)
    if a.sustained:
        print(json.dumps(run_sustained(ctx, a.width, a.height, 10, a.frames, a.threads or None, a.tile_cols, a.tile_rows, depth=a.sustained)))
        return
    print(json.dumps(run(ctx, a.width, a.height, 10, a.frames, a.threads or None, a.tile_cols, intra_pct=a.intra_pct, key_frame=bool(a.key_frame), tile_rows=a.tile_rows)))


if __name__ == "__main__":
    main()
