"""dav1d's own task loop with the backend plugged in (oracle/ref_hooked.c inside oracle/_ref_hooked/libdav1d_hooked.so — the reference
build with src/thread_task.c patched at the hook points of INTEGRATION.md 2).  TEST INFRASTRUCTURE: the product never loads this."""
import ctypes as C
import os

import numpy as np

import util
import lister_util as lu
from dav1d_amd import _lib

HOOKED_SO = os.path.join(util.ROOT, "oracle", "_ref_hooked", "libdav1d_hooked.so")


class HookedParams(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("layout", C.c_int), ("bpc", C.c_int), ("sb128", C.c_int),
                ("n_tile_cols", C.c_int), ("n_tile_rows", C.c_int), ("col_start_sb", C.c_uint16 * 65), ("row_start_sb", C.c_uint16 * 65),
                ("n_threads", C.c_int), ("frame_delay", C.c_int), ("n_frames", C.c_int),
                ("lf_level_y", C.c_int * 2), ("lf_level_u", C.c_int), ("lf_level_v", C.c_int), ("lf_sharpness", C.c_int),
                ("cdef_enabled", C.c_int), ("cdef_damping", C.c_int), ("cdef_n_bits", C.c_int), ("cdef_y_strength", C.c_int * 8),
                ("cdef_uv_strength", C.c_int * 8), ("lr_type", C.c_int * 3), ("lr_unit_size", C.c_int * 2),
                ("mode", C.c_int), ("free_listing", C.c_int), ("device", C.c_int), ("keep_output", C.c_int), ("synth", _lib.SynthParams)]


def lib():
    if not os.path.exists(HOOKED_SO):
        return None
    l = C.CDLL(HOOKED_SO)
    l.dav1d_hooked_open.restype = C.c_void_p
    l.dav1d_hooked_open.argtypes = [C.POINTER(HookedParams), C.c_char_p]
    l.dav1d_hooked_run.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    l.dav1d_hooked_plane.restype = C.c_void_p
    l.dav1d_hooked_plane.argtypes = [C.c_void_p, C.c_int, C.c_int]
    l.dav1d_hooked_n_fc.argtypes = [C.c_void_p]
    l.dav1d_hooked_close.argtypes = [C.c_void_p]
    return l


FILTERS = dict(lf=(20, 28, 16, 24, 0), cdef=(5, 2, [17, 33, 0, 63], [5, 0, 20, 48]), lr=([1, 1, 1], [6, 6]))


def params(w, h, bpc, n_frames, mode, layout=1, sb128=True, tiles=(2, 1), threads=4, frame_delay=3, filters=FILTERS, seed=5, free_listing=1,
           keep_output=True, synth=None):
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
    p.synth = synth if synth is not None else lu.default_synth(seed, n_refs=3, far_mv_pct=2)
    return p


def run(p, hip_lib_path):
    """One chain through dav1d's task loop; returns (seconds, n_fc, [frame][plane] arrays or None)."""
    l = lib()
    assert l is not None, "oracle/_ref_hooked is not built"
    h = l.dav1d_hooked_open(C.byref(p), hip_lib_path.encode())
    assert h, "dav1d_hooked_open failed"
    try:
        sec = C.c_double()
        rc = l.dav1d_hooked_run(h, C.byref(sec))
        assert rc == 0, "dav1d_hooked_run: %d" % rc
        n_fc = l.dav1d_hooked_n_fc(h)
        frames = None
        if p.keep_output:
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
