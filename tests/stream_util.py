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
    l.dav1d_hooked_stream_tile_errors.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.c_int]
    l.dav1d_hooked_stream_histogram.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    return l


def decode(units, mode, hip_lib_path, threads=4, frame_delay=3, free_listing=1, pack=True, apply_grain=True, keep=True):
    """units: list of bytes-like temporal units.  Returns dict(pictures=[(info, [planes])], errors=n, tile_errors=[(tu, offset, overread)],
    hist={...} (mode 1), seconds)."""
    l = lib()
    assert l is not None, "oracle/_ref_hooked is not built"
    p = hk.HookedParams()
    p.n_threads, p.frame_delay, p.n_frames = threads, frame_delay, 0
    p.mode, p.free_listing, p.device, p.keep_output = mode, free_listing, 0, int(keep)
    p.pack = int(bool(pack) and mode == 1)
    p.stream, p.apply_grain = 1, int(apply_grain)
    h = l.dav1d_hooked_open(C.byref(p), hip_lib_path.encode(), None)
    assert h, "dav1d_hooked_open failed"
    try:
        bufs = [(C.c_uint8 * len(u)).from_buffer_copy(bytes(u)) for u in units]
        ptrs = (C.c_void_p * len(bufs))(*[C.addressof(b) for b in bufs])
        sizes = (C.c_size_t * len(bufs))(*[len(b) for b in bufs])
        sec = C.c_double()
        rc = l.dav1d_hooked_stream_run(h, ptrs, sizes, len(bufs), C.byref(sec))
        assert rc == 0, "dav1d_hooked_stream_run: %d (the backend failed)" % rc
        n = l.dav1d_hooked_stream_pictures(h)
        pics = []
        for i in range(n):
            info = (C.c_int * 6)()
            planes = []
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
        return dict(pictures=pics, errors=l.dav1d_hooked_stream_errors(h), tile_errors=[(te[3 * i], te[3 * i + 1], te[3 * i + 2]) for i in range(nte)],
                    hist=dict(zip(HIST, [int(v) for v in hist[:nh]])), seconds=sec.value)
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
