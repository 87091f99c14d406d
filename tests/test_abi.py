"""C-ABI surface: the built library loads, exports every symbol include/dav1d_hip.h declares,
agrees with the Python struct mirrors, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import util
from dav1d_amd import _lib, build


@pytest.fixture(scope="module")
def so():
    return build.build_hip()


def test_header_symbols_are_all_exported(so):
    hdr = open(os.path.join(util.ROOT, "include", "dav1d_hip.h")).read()
    declared = set(re.findall(r"DAV1D_HIP_API\s+[\w\s\*]+?\b(dav1d_hip_\w+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = C.CDLL(so)
    for s in declared:
        assert hasattr(lib, s), s


def test_struct_mirrors_match_header_sizes():
    assert _lib.ITX_TASK.itemsize == 16
    assert _lib.MC_TASK.itemsize == 24
    assert _lib.COMP_TASK.itemsize == 24
    assert C.sizeof(_lib.Plane) == 24 and C.sizeof(_lib.Picture) == 24 * 3 + 8 + 16 + 3 * 8 + 8 + 8      # planes, bpc + layout, alloc, twin[3], twin_alloc, twin_ok (+ padding)


def test_open_fails_loudly_without_device(so):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load(so)
    h = C.c_void_p()
    assert lib.dav1d_hip_open(C.byref(h), 0, None) == -19     # -ENODEV
    tab = (C.c_void_p * 1024)()
    assert lib.dav1d_hip_dsp_init_16bpc(tab, 10) == -19
    assert not any(tab)
    from dav1d_amd import api
    with pytest.raises(api.HipError):
        api.Context(0)


def test_missing_library_raises():
    with pytest.raises(_lib.LibraryError):
        _lib.load("/nonexistent/libdav1d_hip.so")


def test_device_tables_equal_reference_tables():
    """dav1d_amd/csrc/av1_tables.h (generated) vs the tables of the reference build."""
    lib = util.ref_lib()
    if lib is None:
        pytest.skip("oracle/_ref not built")
    txt = open(os.path.join(util.ROOT, "dav1d_amd", "csrc", "av1_tables.h")).read()
    for m in re.finditer(r"AV1_TABLE_QUAL (\w+) av1_(\w+)\[(\d+)\] = \{[^\n]*\n(.*?)\};", txt, re.S):
        ctype, name, n, body = m.group(1), m.group(2), int(m.group(3)), m.group(4)
        if name in ("mc_taps_packed", "mc_tap_span"):
            continue                      # derived table, checked in test_packed_taps_match_subpel_filters
        vals = np.array([int(v) for v in body.replace("\n", " ").split(",") if v.strip()])
        dt = {"int8_t": np.int8, "uint8_t": np.uint8, "int16_t": np.int16, "uint16_t": np.uint16}[ctype]
        sz = C.c_size_t()
        p = lib.dav1d_ref_table(name.encode(), C.byref(sz))
        assert p and sz.value == n * np.dtype(dt).itemsize, name
        ref = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(sz.value,)).view(dt)
        assert np.array_equal(ref, vals.astype(dt)), name


def test_packed_taps_match_subpel_filters():
    """av1_mc_taps_packed (v_dot2 operand form) is a pure re-packing of av1_mc_subpel_filters; phase 0 is the unit tap at the
    filters' own scale (64, bilinear 16: tools/gen_tables.py)."""
    txt = open(os.path.join(util.ROOT, "dav1d_amd", "csrc", "av1_tables.h")).read()

    def table(name):
        body = re.search(r"av1_%s\[\d+\] = \{[^\n]*\n(.*?)\};" % name, txt, re.S).group(1)
        return [int(v.strip().rstrip("u"), 0) for v in body.replace("\n", " ").split(",") if v.strip()]

    filt = np.array(table("mc_subpel_filters")).reshape(6, 15, 8)
    packed = np.array(table("mc_taps_packed"), dtype=np.uint32).reshape(7, 16, 9)
    for fs in range(7):
        for m in range(16):
            f = [0, 0, 0, 16 if fs == 6 else 64, 0, 0, 0, 0] if m == 0 else ([0, 0, 0, 16 - m, m, 0, 0, 0] if fs == 6 else list(filt[fs, m - 1]))
            g = [0] + f + [0]
            want = [(f[2 * k] & 0xffff) | ((f[2 * k + 1] & 0xffff) << 16) for k in range(4)] + \
                   [(g[2 * k] & 0xffff) | ((g[2 * k + 1] & 0xffff) << 16) for k in range(5)]
            assert list(packed[fs, m]) == want, (fs, m)
    span = np.array(table("mc_tap_span")).reshape(7, 16)
    for fs in range(7):
        for m in range(16):
            f = [0, 0, 0, 16 if fs == 6 else 64, 0, 0, 0, 0] if m == 0 else ([0, 0, 0, 16 - m, m, 0, 0, 0] if fs == 6 else list(filt[fs, m - 1]))
            lo, hi = span[fs, m] & 15, span[fs, m] >> 4
            assert all(v == 0 for v in f[:lo]) and all(v == 0 for v in f[hi:]) and f[lo] and f[hi - 1], (fs, m)
