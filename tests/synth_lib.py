"""tests/synth/libdav1d_synth.so: the generator of synthetic pass-1 output (Av1Block / cbi / cf / palettes of one frame drawn from a seeded
generator under the legality rules of the AV1 syntax).  TEST INFRASTRUCTURE — used by the parity tests against the reference's own pass
2, by bench.py's host-side legs and by the chain mode of oracle/ref_hooked.c; the product library neither contains nor loads it.  Frames
decoded by dav1d's REAL pass 1 come from tests/av1_obu.py + tests/stream_util.py instead."""
import ctypes as C
import os
import subprocess

from dav1d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "synth", "libdav1d_synth.so")


class SynthParams(C.Structure):        # == Dav1dSynthParams, tests/synth/dav1d_synth.h
    _fields_ = [("seed", C.c_uint64), ("intra_pct", C.c_int), ("skip_pct", C.c_int), ("compound_pct", C.c_int),
                ("masked_compound", C.c_int), ("global_pct", C.c_int), ("interintra_pct", C.c_int), ("obmc_pct", C.c_int),
                ("warp_pct", C.c_int), ("cfl_pct", C.c_int), ("palette", C.c_int), ("filter_intra_pct", C.c_int),
                ("tx_split_pct", C.c_int), ("alt_txtp_pct", C.c_int), ("eob_none_pct", C.c_int), ("mv_range", C.c_int),
                ("far_mv_pct", C.c_int), ("n_refs", C.c_int), ("split_pct", C.c_int * 5), ("rect_pct", C.c_int),
                ("fixed_bl", C.c_int), ("cf_align64", C.c_int), ("intrabc_pct", C.c_int), ("n_segs", C.c_int),
                ("skip_mode_pct", C.c_int)]


_cached = None


def lib():
    global _cached
    if _cached is None:
        if not os.path.exists(PATH):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "synth"], check=True, stdout=subprocess.DEVNULL)
        l = C.CDLL(PATH)
        l.dav1d_synth_frame.restype = C.c_int
        l.dav1d_synth_frame.argtypes = [C.POINTER(_lib.FrameDesc), C.POINTER(SynthParams), C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        _cached = l
    return _cached


def synth_frame(desc, sp, cf, cf_bytes, cbi_entries, pal_idx, pal_idx_bytes):
    return lib().dav1d_synth_frame(C.byref(desc), C.byref(sp), cf, cf_bytes, cbi_entries, pal_idx, pal_idx_bytes)
