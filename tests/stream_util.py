"""AV1 streams (tests/av1_obu.py: real headers, random tile payloads) through dav1d's public API inside oracle/_ref_hooked — the reference's
own dav1d_parse_obus / dav1d_submit_frame / msac / decode_b / task loop — with and without the backend plugged in.  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

import av1_obu
import hooked_util as hk

HIST = ["frames_key", "frames_inter", "frames_intra_only", "frames_super_res", "frames_scaled_refs", "frames_intrabc", "frames_film_grain",
        "frames_delta_lf", "frames_segmented", "frames_lossless",
        "b_intra", "b_inter", "b_intrabc", "b_skip", "b_skip_mode", "b_seg_nonzero",
        "b_palette_y", "b_palette_uv", "b_cfl", "b_filter_intra", "b_directional", "b_angle_delta", "b_smooth", "b_paeth",
        "b_comp_avg", "b_comp_wavg", "b_comp_seg", "b_comp_wedge", "b_interintra", "b_interintra_wedge", "b_obmc",
        "b_local_warp", "b_globalmv", "b_global_warp", "b_scaled_ref", "b_filter_not_regular", "b_dual_filter",
        "b_tx_split", "b_tx64", "b_lossless", "b_sub8x8_chroma", "b_128", "b_4xn",
        "tx_blocks", "tx_non_dct", "tx_eob0", "tx_no_coefs", "lr_wiener", "lr_sgr", "cdef_nonzero_idx"]


def lib():
    l = hk.lib()
    if l is None or not hasattr(l, "dav1d_hooked_stream_run"):
        return None
    l.dav1d_hooked_stream_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_double)]
    l.dav1d_hooked_stream_pictures.argtypes = [C.c_void_p]
    l.dav1d_hooked_stream_errors.argtypes = [C.c_void_p]
    l.dav1d_hooked_stream_picture.restype = C.c_void_p
    l.dav1d_hooked_stream_picture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    l.dav1d_hooked_stream_picture_digest.restype = C.c_double
    l.dav1d_hooked_stream_picture_digest.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    l.dav1d_hooked_stream_tile_errors.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.c_int]
    l.dav1d_hooked_stream_histogram.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    return l


def decode(units, mode, hip_lib_path, threads=4, frame_delay=3, free_listing=1, pack=True, apply_grain=True, keep=True, row_progress=0,
           filters_off=0, allow_backend_failure=False, n_devices=0):
    """units: list of bytes-like temporal units.  Returns dict(pictures=[(info, [planes])], errors=n, tile_errors=[(tu, offset, overread)],
    hist={...} (mode 1), seconds, digests=[(d0, d1, d2)], times=[...]).  keep = 2: digests of the planes only (no pixel copies kept)."""
    l = lib()
    assert l is not None, "oracle/_ref_hooked is not built"
    p = hk.HookedParams()
    p.n_threads, p.frame_delay, p.n_frames = threads, frame_delay, 0
    p.mode, p.free_listing, p.device, p.keep_output = mode, free_listing, 0, int(keep)
    p.pack = int(bool(pack) and mode == 1)
    p.stream, p.apply_grain, p.row_progress = 1, int(apply_grain), int(row_progress)
    p.filters_off = int(filters_off)         # Dav1dSettings.inloop_filters = ALL & ~filters_off, both modes
    p.n_devices = int(n_devices)             # mode 1: the binding ends frames on that many devices in turn
    h = l.dav1d_hooked_open(C.byref(p), hip_lib_path.encode(), None)
    assert h, "dav1d_hooked_open failed"
    try:
        bufs = [(C.c_uint8 * len(u)).from_buffer_copy(bytes(u)) for u in units]
        ptrs = (C.c_void_p * len(bufs))(*[C.addressof(b) for b in bufs])
        sizes = (C.c_size_t * len(bufs))(*[len(b) for b in bufs])
        sec = C.c_double()
        rc = l.dav1d_hooked_stream_run(h, ptrs, sizes, len(bufs), C.byref(sec))
        assert rc == 0 or allow_backend_failure, "dav1d_hooked_stream_run: %d (the backend failed)" % rc
        n = l.dav1d_hooked_stream_pictures(h)
        pics, digests, times = [], [], []
        for i in range(n):
            info = (C.c_int * 6)()
            planes = []
            dg = (C.c_uint64 * 3)()
            times.append(l.dav1d_hooked_stream_picture_digest(h, i, dg))
            digests.append(tuple(int(v) for v in dg))
            if int(keep) == 2:
                l.dav1d_hooked_stream_picture(h, i, 0, info)
                pics.append((dict(w=info[0], h=info[1], layout=info[2], bpc=info[3], frame_offset=info[4], grain=info[5]), []))
                continue
            for pl in range(3):
                ptr = l.dav1d_hooked_stream_picture(h, i, pl, info)
                if not ptr:
                    break
                w, hh, layout, bpc = info[0], info[1], info[2], info[3]
                ss_hor, ss_ver = int(layout != 3), int(layout == 1)
                pw = w if not pl else (w + ss_hor) >> ss_hor
                ph = hh if not pl else (hh + ss_ver) >> ss_ver
                dt = np.uint8 if bpc == 8 else np.uint16
                planes.append(np.ctypeslib.as_array((C.c_uint8 * (pw * ph * dt().itemsize)).from_address(ptr)).view(dt).reshape(ph, pw).copy())
            pics.append((dict(w=info[0], h=info[1], layout=info[2], bpc=info[3], frame_offset=info[4], grain=info[5]), planes))
        te = (C.c_long * (3 * 64))()
        nte = min(64, l.dav1d_hooked_stream_tile_errors(h, te, 64))
        hist = (C.c_uint64 * 64)()
        nh = l.dav1d_hooked_stream_histogram(h, hist, 64)
        assert nh == len(HIST), (nh, len(HIST))
        st = (C.c_double * 16)()
        l.dav1d_hooked_stats(h, st)
        stats = dict(zip(("picture_alloc", "after_init", "listing", "filter_listing", "gpu_thread_idle", "uploads", "frame_end", "fetch", "picture_release"),
                         [round(v * 1e3 / max(1, n), 2) for v in st[:9]]))
        dev_stats = []
        if mode == 1:
            ds = (C.c_int * 2)()
            for d in range(max(1, hk.lib().dav1d_hooked_device_stats(h, 0, C.byref(ds)))):
                hk.lib().dav1d_hooked_device_stats(h, d, C.byref(ds))
                dev_stats.append((int(ds[0]), int(ds[1])))
        return dict(rc=rc, device_stats=dev_stats, row_publications=l.dav1d_hooked_row_publications(h), stats=stats, pictures=pics, errors=l.dav1d_hooked_stream_errors(h), tile_errors=[(te[3 * i], te[3 * i + 1], te[3 * i + 2]) for i in range(nte)],
                    hist=dict(zip(HIST, [int(v) for v in hist[:nh]])), seconds=sec.value, digests=digests, times=times)
    finally:
        l.dav1d_hooked_close(h)


def repair(sw, hip_lib_path, threads=4, max_rounds=400, verbose=False):
    """Re-rolls tile payloads dav1d's pass 1 rejects (4:2:2 forbids some partitions, an intra block copy may find no source,
    src/decode.c:2150-2156, 1337-1341) from a few bytes before the point the symbol decoder had reached, earliest temporal unit
    first, until the whole stream decodes.  Returns the number of decodes it took."""
    for rnd in range(max_rounds):
        units = [u["data"] for u in sw.units]
        r = decode(units, 0, hip_lib_path, threads=threads, keep=False)
        if not r["tile_errors"] and not r["errors"]:
            return rnd + 1
        assert r["tile_errors"], "dav1d rejects the stream outside the tile data (%d errors): a header the writer gets wrong" % r["errors"]
        first = min(t[0] for t in r["tile_errors"])
        for tu, off, over in r["tile_errors"]:
            if tu != first:
                continue
            u = sw.units[tu]
            hit = [i for i, (o, s) in enumerate(u["tiles"]) if o <= off <= o + s]
            assert hit, (tu, off, u["tiles"])
            ti = hit[0]
            assert not over, "tile %d of unit %d ran out of data: raise Knobs.bytes_per_pixel" % (ti, tu)
            rel = off - u["tiles"][ti][0]
            if verbose:
                print("unit %d tile %d: re-roll from byte %d of %d" % (tu, ti, rel - 10, u["tiles"][ti][1]))
            sw.reroll(tu, ti, rel - 10)
    raise AssertionError("stream still rejected after %d rounds" % max_rounds)


def compare(want, got):
    """picture by picture; returns a description of the first difference or None"""
    if len(want["pictures"]) != len(got["pictures"]):
        return "%d pictures against %d" % (len(got["pictures"]), len(want["pictures"]))
    for i, ((wi, wp), (gi, gp)) in enumerate(zip(want["pictures"], got["pictures"])):
        if wi != gi:
            return "picture %d: %s against %s" % (i, gi, wi)
        for pl in range(len(wp)):
            if not np.array_equal(wp[pl], gp[pl]):
                bad = np.argwhere(wp[pl] != gp[pl])
                return ("picture %d (order hint %d, %dx%d) plane %d: %d pixels differ, first at (y, x) = %s, rows %d..%d, columns %d..%d, want %d got %d"
                        % (i, wi["frame_offset"], wi["w"], wi["h"], pl, len(bad), tuple(bad[0]), bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(),
                           bad[:, 1].max(), wp[pl][tuple(bad[0])], gp[pl][tuple(bad[0])]))
    return None


def task_loop_rate(hip_lib_path, w, h, bpc, tiles_log2=(2, 0), threads=64, frame_delay=8, frames=24, seed=0x57EA, bytes_per_pixel=0.1, seg_pin=0):
    """bench.py's dav1d_task_loop leg: an AV1 stream of the BASELINE configs[2] geometry (8K 4:2:0 10-bit, 4 tile columns) — a key
    frame, then inter frames with every tool and filter on (film grain included; frame sizes kept constant so that the rate means
    something), tile payloads random — decoded by dav1d alone (the peer: its C pass 2 + filters + grain on the same worker threads)
    and with the backend behind dav1d's real pass 1.  EVERY picture of the chain is compared (plane digests)."""
    k = av1_obu.Knobs(hidden=0.0, super_res=0.0, scaled=0.0, intra_only=0.0, non_uniform_tiles=0.0, lossless=0.0, bytes_per_pixel=bytes_per_pixel,
                      screen_content=0.15, error_resilient=0.0, seg_pin=seg_pin)
    seq = av1_obu.Seq(w, h, av1_obu.LAYOUT_I420, bpc, True)
    sw = av1_obu.StreamWriter(seq, seed, k)
    while sw.n < frames:
        sw.next_frame(tiles_log2=tiles_log2)
    rounds = repair(sw, hip_lib_path, threads=threads, max_rounds=50)
    units = [u["data"] for u in sw.units]
    peer = decode(units, 0, hip_lib_path, threads=threads, frame_delay=frame_delay, keep=2)
    got = decode(units, 1, hip_lib_path, threads=threads, frame_delay=frame_delay, keep=2)
    assert len(peer["digests"]) == len(got["digests"]) == frames and not peer["errors"] and not got["errors"], (len(peer["digests"]), len(got["digests"]))
    bad = [i for i in range(frames) if peer["digests"][i] != got["digests"][i]]
    assert not bad, "dav1d task loop behind dav1d's real pass 1: pictures %s differ from dav1d's own" % bad

    def steady(r, skip):
        t = r["times"]
        n = len(t) - 1 - skip
        dt = t[-1] - t[skip]
        return dict(frames=n, fps=round(n / dt, 2), ms_per_frame=round(dt / n * 1e3, 2), value=round(w * h * n / dt / 1e6, 1), unit="Mpixels/s")
    skip = min(frame_delay, frames - 2)
    fr = [u["frame"] for u in sw.units if u["frame"] is not None]
    return {"frames": frames, "fps": round(frames / got["seconds"], 2), "ms_per_frame": round(got["seconds"] / frames * 1e3, 2),
            "value": round(w * h * frames / got["seconds"] / 1e6, 1), "unit": "Mpixels/s",
            "steady_state": steady(got, skip), "peer_steady_state": steady(peer, skip), "peer_fps": round(frames / peer["seconds"], 2),
            "peer": "dav1d itself on the same stream: the reference's pass 2 + in-loop filters + film grain (C, no assembly: no nasm on the box), "
                    "%d worker threads, same %d frames, steady state timed over the same pictures" % (threads, frames),
            "parity": "bit-exact vs dav1d on ALL %d pictures (plane digests)" % frames, "worker_threads": threads, "frame_delay": frame_delay,
            "ms_per_frame_by_stage_summed_over_threads": got["stats"],
            "segments_pinned_to_a_reference": seg_pin,
            "tile_cols": 1 << tiles_log2[0], "tile_rows": 1 << tiles_log2[1], "stream_bytes": sum(len(u) for u in units), "repair_rounds": rounds,
            "tools_in_the_stream": {kk: v for kk, v in got["hist"].items() if v},
            "frames_with_film_grain": sum(1 for f in fr if f.grain is not None),
            "workload": "%dx%d 4:2:0 %d-bit AV1 stream written by tests/av1_obu.py (real sequence / frame headers, every tool on, random tile payloads): "
                        "dav1d_send_data / dav1d_parse_obus / dav1d_submit_frame / msac / decode_b unmodified, pass 1 on dav1d's workers; "
                        "the backend behind it through the hook points of INTEGRATION.md 2; film grain applied to every output picture" % (w, h, bpc)}
