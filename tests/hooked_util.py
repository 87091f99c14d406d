"""dav1d's own task loop with the backend plugged in (oracle/ref_hooked.c inside oracle/_ref_hooked/libdav1d_hooked.so — the reference
build with src/thread_task.c patched at the hook points of INTEGRATION.md 2).  TEST INFRASTRUCTURE: the product never loads this."""
import ctypes as C
import os

import numpy as np

import util
import lister_util as lu
import synth_lib
from dav1d_amd import _lib

HOOKED_SO = os.path.join(util.ROOT, "oracle", "_ref_hooked_release" if util.REF_RELEASE else "_ref_hooked", "libdav1d_hooked.so")


class HookedParams(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("layout", C.c_int), ("bpc", C.c_int), ("sb128", C.c_int),
                ("n_tile_cols", C.c_int), ("n_tile_rows", C.c_int), ("col_start_sb", C.c_uint16 * 65), ("row_start_sb", C.c_uint16 * 65),
                ("n_threads", C.c_int), ("frame_delay", C.c_int), ("n_frames", C.c_int),
                ("lf_level_y", C.c_int * 2), ("lf_level_u", C.c_int), ("lf_level_v", C.c_int), ("lf_sharpness", C.c_int),
                ("cdef_enabled", C.c_int), ("cdef_damping", C.c_int), ("cdef_n_bits", C.c_int), ("cdef_y_strength", C.c_int * 8),
                ("cdef_uv_strength", C.c_int * 8), ("lr_type", C.c_int * 3), ("lr_unit_size", C.c_int * 2),
                ("mode", C.c_int), ("free_listing", C.c_int), ("device", C.c_int), ("keep_output", C.c_int), ("inject", C.c_int), ("pack", C.c_int),
                ("synth", synth_lib.SynthParams), ("stream", C.c_int), ("row_progress", C.c_int), ("apply_grain", C.c_int), ("filters_off", C.c_int),
                ("n_devices", C.c_int)]


def lib():
    if not os.path.exists(HOOKED_SO):
        return None
    l = C.CDLL(HOOKED_SO)
    l.dav1d_hooked_open.restype = C.c_void_p
    l.dav1d_hooked_open.argtypes = [C.POINTER(HookedParams), C.c_char_p, C.c_void_p]
    l.dav1d_hooked_store_create.restype = C.c_void_p
    l.dav1d_hooked_store_create.argtypes = [C.c_int]
    l.dav1d_hooked_store_destroy.argtypes = [C.c_void_p]
    l.dav1d_hooked_run.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.dav1d_hooked_plane.restype = C.c_void_p
    l.dav1d_hooked_plane.argtypes = [C.c_void_p, C.c_int, C.c_int]
    l.dav1d_hooked_n_fc.argtypes = [C.c_void_p]
    l.dav1d_hooked_row_publications.argtypes = [C.c_void_p]
    l.dav1d_hooked_device_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int * 2)]
    if hasattr(l, "dav1d_hooked_band_copies"):
        l.dav1d_hooked_band_copies.argtypes = [C.c_void_p, C.c_int]
    l.dav1d_hooked_tail_seconds.restype = C.c_double
    l.dav1d_hooked_tail_seconds.argtypes = [C.c_void_p, C.c_int]
    l.dav1d_hooked_output_tail_seconds.restype = C.c_double
    l.dav1d_hooked_output_tail_seconds.argtypes = [C.c_void_p, C.c_int]
    l.dav1d_hooked_picture_digest.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    l.dav1d_hooked_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.dav1d_hooked_frame_end_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.dav1d_hooked_close.argtypes = [C.c_void_p]
    return l


FILTERS = dict(lf=(20, 28, 16, 24, 0), cdef=(5, 2, [17, 33, 0, 63], [5, 0, 20, 48]), lr=([1, 1, 1], [6, 6]))


def params(w, h, bpc, n_frames, mode, layout=1, sb128=True, tiles=(2, 1), threads=4, frame_delay=3, filters=FILTERS, seed=5, free_listing=1,
           keep_output=True, synth=None, pack=True, row_progress=0, n_devices=0):
    p = HookedParams()
    p.w, p.h, p.layout, p.bpc, p.sb128 = w, h, layout, bpc, int(sb128)
    sb = 128 if sb128 else 64
    cs = lu.uniform_tiles((w + sb - 1) // sb, tiles[0])
    rs = lu.uniform_tiles((h + sb - 1) // sb, tiles[1])
    p.n_tile_cols, p.n_tile_rows = len(cs) - 1, len(rs) - 1
    for i, v in enumerate(cs):
        p.col_start_sb[i] = v
    for i, v in enumerate(rs):
        p.row_start_sb[i] = v
    p.n_threads, p.frame_delay, p.n_frames = threads, frame_delay, n_frames
    if filters and "lf" in filters:
        p.lf_level_y[0], p.lf_level_y[1], p.lf_level_u, p.lf_level_v, p.lf_sharpness = filters["lf"]
    if filters and "cdef" in filters:
        damping, n_bits, ys, uvs = filters["cdef"]
        p.cdef_enabled, p.cdef_damping, p.cdef_n_bits = 1, damping, n_bits
        for i in range(1 << n_bits):
            p.cdef_y_strength[i], p.cdef_uv_strength[i] = ys[i], uvs[i]
    if filters and "lr" in filters:
        for i in range(3):
            p.lr_type[i] = filters["lr"][0][i]
        p.lr_unit_size[0], p.lr_unit_size[1] = filters["lr"][1]
    p.mode, p.free_listing, p.device, p.keep_output = mode, free_listing, 0, int(keep_output)
    p.row_progress = int(row_progress)
    p.n_devices = int(n_devices)
    p.pack = int(bool(pack) and mode == 1 and os.environ.get("DAV1D_HOOKED_PACK", "1") != "0")
    p.synth = synth if synth is not None else lu.default_synth(seed, n_refs=3, far_mv_pct=2)
    return p


class Store:
    """pass 1's output of the frames of a chain, kept between runs (HookedParams.inject: 1 fills it, 2 replays it)"""

    def __init__(self, n_frames):
        self.l = lib()
        self.h = self.l.dav1d_hooked_store_create(n_frames)
        assert self.h

    def destroy(self):
        if self.h:
            self.l.dav1d_hooked_store_destroy(self.h)
            self.h = None


def run(p, hip_lib_path, store=None, inject=0):
    """One chain through dav1d's task loop; returns (seconds, n_fc, [frame][plane] arrays or None)."""
    l = lib()
    assert l is not None, "oracle/_ref_hooked is not built"
    p.inject = inject if store is not None else 0
    h = l.dav1d_hooked_open(C.byref(p), hip_lib_path.encode(), store.h if store is not None else None)
    assert h, "dav1d_hooked_open failed"
    try:
        sec = C.c_double()
        rc = l.dav1d_hooked_run(h, C.byref(sec))
        assert rc == 0, "dav1d_hooked_run: %d" % rc
        n_fc = l.dav1d_hooked_n_fc(h)
        run.last_row_publications = l.dav1d_hooked_row_publications(h)
        run.last_device_stats = []          # mode 1, per device of the binding: (frames ended there, reference pictures copied there)
        if p.mode == 1:
            ds = (C.c_int * 2)()
            for d in range(max(1, l.dav1d_hooked_device_stats(h, 0, C.byref(ds)))):
                l.dav1d_hooked_device_stats(h, d, C.byref(ds))
                run.last_device_stats.append((int(ds[0]), int(ds[1])))
            run.last_band_copies = [int(l.dav1d_hooked_band_copies(h, d)) for d in range(len(run.last_device_stats))] if hasattr(l, "dav1d_hooked_band_copies") else []
        st = (C.c_double * 16)()
        l.dav1d_hooked_stats(h, st)
        run.last_stats = dict(zip(("picture_alloc", "after_init", "listing", "filter_listing", "gpu_thread_idle", "uploads", "frame_end", "fetch", "picture_release"),
                                  [round(v * 1e3 / max(1, p.n_frames), 2) for v in st[:9]]))
        run.last_tail = (lambda frm: l.dav1d_hooked_tail_seconds(h, frm))(getattr(run, "tail_from", 0)) if p.mode == 1 else 0.
        run.last_out_tail = l.dav1d_hooked_output_tail_seconds(h, getattr(run, "tail_from", 0))
        run.last_digests = None
        if p.keep_output == 2:
            run.last_digests = []
            for k in range(p.n_frames):
                dg = (C.c_uint64 * 3)()
                assert l.dav1d_hooked_picture_digest(h, k, dg) == 0
                run.last_digests.append(tuple(int(v) for v in dg))
        fe = (C.c_double * 64)()
        l.dav1d_hooked_frame_end_seconds(h, fe)
        run.last_frame_end_ms = [round(v * 1e3, 2) for v in fe[:min(64, p.n_frames)]]
        frames = None
        if p.keep_output == 1:
            dt = np.uint8 if p.bpc == 8 else np.uint16
            ss_hor, ss_ver = int(p.layout != 3), int(p.layout == 1)
            frames = []
            for k in range(p.n_frames):
                planes = []
                for pl in range(1 if p.layout == 0 else 3):
                    w = p.w if not pl else (p.w + ss_hor) >> ss_hor
                    hh = p.h if not pl else (p.h + ss_ver) >> ss_ver
                    ptr = l.dav1d_hooked_plane(h, k, pl)
                    assert ptr, (k, pl)
                    planes.append(np.ctypeslib.as_array((C.c_uint8 * (w * hh * dt().itemsize)).from_address(ptr)).view(dt).reshape(hh, w).copy())
                frames.append(planes)
        return sec.value, n_fc, frames
    finally:
        l.dav1d_hooked_close(h)


def task_loop_rate(hip_lib_path, w, h, bpc, tiles=(4, 1), threads=64, frame_delay=8, frames=24, check_frames=4, seed=0x7A5C, intra_pct=10, n_devices=0):
    """bench.py's dav1d_task_loop leg: a chain of dependent frames (a key frame, then inter frames of the C2 block mix predicting from
    the three frames before them; deblocking, CDEF and switchable restoration on) through dav1d's OWN task loop — dav1d_submit_frame,
    its worker threads, check_tile, dav1d_get_picture — with the backend plugged in at the hook points of INTEGRATION.md 2.  First a
    short chain is decoded both ways and compared picture by picture (the peer: the reference's pass 2 + filters on the same worker
    threads, C only); then the chain is timed, first frame submitted to last picture out."""
    import e2e
    sp = e2e.c2_params(seed)
    sp.intra_pct = intra_pct
    common = dict(tiles=tiles, threads=threads, frame_delay=frame_delay, synth=sp, n_devices=n_devices)      # (n_devices: the binding's, mode 1 only)
    store = Store(frames)
    try:
        # the peer generates (and keeps) pass 1's output of EVERY frame of the chain and leaves the digests of its pictures; everything after
        # replays the store, so that the generator and the mask-building walk (one thread per frame in this harness) stay out of the timed chains
        run.tail_from = min(frame_delay, frames - 2)
        _, n_fc, _ = run(params(w, h, bpc, frames, mode=0, keep_output=2, **common), hip_lib_path, store, inject=1)
        want = list(run.last_digests)
        run(params(w, h, bpc, frames, mode=1, keep_output=2, **common), hip_lib_path, store, inject=2)
        got = list(run.last_digests)
        bad = [k for k in range(frames) if want[k] != got[k]]
        assert not bad, "dav1d task loop: frames %s of %d differ from dav1d's own pass 2 + filters (plane digests)" % (bad, frames)
        # the peer timed over the same chain, steady state over the same frames (replaying the store: copies, its pass 2 consumes cf)
        cpu_s, _, _ = run(params(w, h, bpc, frames, mode=0, keep_output=False, **common), hip_lib_path, store, inject=2)
        peer_tail_s = run.last_out_tail
        # last: the packing lister consumes the store's coefficient arrays (as dav1d's pass 2 consumes f->frame_thread.cf)
        t_s, _, _ = run(params(w, h, bpc, frames, mode=1, keep_output=False, **common), hip_lib_path, store, inject=2)
        tail_s, tail_n = run.last_tail, frames - 1 - run.tail_from
        dev_stats = list(run.last_device_stats)
        stages = dict(run.last_stats)
        stages["frame_end_ms_by_frame"] = list(run.last_frame_end_ms)
        # the same chain once more with rows published to dav1d's progress[1] as the backend reports them (C callback: dav1d_hooked_rows_done)
        fe_plain = list(run.last_frame_end_ms)
        rp = None
        try:
            run.tail_from = min(frame_delay, frames - 2)
            run(params(w, h, bpc, frames, mode=1, keep_output=False, row_progress=1, **common), hip_lib_path, store, inject=2)
            fe_rows = list(run.last_frame_end_ms)
            k0 = min(frame_delay, frames - 2) + 1
            med = lambda v: float(np.median(v[k0:])) if len(v) > k0 else None
            a_ms, b_ms = med(fe_plain), med(fe_rows)
            rp = {"steady_state_fps": round((frames - 1 - run.tail_from) / run.last_tail, 1) if run.last_tail else None,
                  "frame_end_ms_median": b_ms, "frame_end_ms_median_without": a_ms,
                  "cost_pct_of_frame_end": round((b_ms / a_ms - 1) * 100, 1) if a_ms and b_ms else None,
                  "rows_published": int(getattr(run, "last_row_publications", 0))}
        except Exception as e:          # noqa: BLE001 - the leg reports, the main figure stands
            rp = {"error": str(e)[:200]}
    finally:
        store.destroy()
    return {"frames": frames, "fps": round(frames / t_s, 1), "ms_per_frame": round(t_s / frames * 1e3, 2),
            "value": round(w * h * frames / t_s / 1e6, 1), "unit": "Mpixels/s",
            "steady_state": {"frames": tail_n, "fps": round(tail_n / tail_s, 1) if tail_s else None, "ms_per_frame": round(tail_s / tail_n * 1e3, 2) if tail_s else None,
                             "value": round(w * h * tail_n / tail_s / 1e6, 1) if tail_s else None,
                             "what": "the inter frames after the first %d (key frame, first-use allocations and pipeline fill left out): completion of frame %d to "
                                     "completion of the last" % (run.tail_from + 1, run.tail_from)}, "n_fc": n_fc, "worker_threads": threads,
            "tile_cols": tiles[0], "tile_rows": tiles[1], "ms_per_frame_by_stage_summed_over_threads": stages, "row_progress": rp,
            "devices": {"n": max(1, n_devices), "frames_ended_and_reference_pictures_copied_in_by_device": dev_stats},
            "peer_fps": round(frames / cpu_s, 2),
            "peer_steady_state": {"frames": tail_n, "fps": round(tail_n / peer_tail_s, 2) if peer_tail_s else None,
                                  "ms_per_frame": round(peer_tail_s / tail_n * 1e3, 2) if peer_tail_s else None},
            "peer": "the reference's pass 2 + in-loop filters (C, no assembly) under the same task loop, %d worker threads, the same %d frames, "
                    "steady state over the same pictures" % (threads, frames),
            "parity": "bit-exact vs dav1d's own pass 2 + filters under the same task loop on ALL %d pictures (plane digests)" % frames,
            "workload": "%dx%d 4:2:0 %d-bit: key frame + inter frames (C2 mix, 10 %% intra, 3 references = the 3 frames before), deblock + CDEF + switchable "
                        "restoration; pass 1's output injected from memory (no AV1 streams exist here); dav1d_open(n_threads=%d, max_frame_delay=%d), src/thread_task.c with the "
                        "hook points of INTEGRATION.md 2; listing runs ahead, %s" % (w, h, bpc, threads, frame_delay, "frames end in order on one GPU thread" if n_devices < 2 else
                        "frame k ends on device k mod %d once its references have ended, a reference of another device copied over first (xGMI)" % n_devices)}
