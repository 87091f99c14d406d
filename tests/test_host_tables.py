"""The lister's AV1 geometry is derived by rule (dav1d_amd/host/av1_host.c); here every derived value is checked against
the tables and functions of the reference build: block / transform dimensions, the largest transform of a block per plane,
partition shapes, all wedge and inter-intra masks, and the warp set-up of MM_WARP blocks."""
import ctypes as C

import numpy as np
import pytest

import util
from dav1d_amd import _lib

ref = util.ref_lib()
pytestmark = pytest.mark.skipif(ref is None, reason="needs the reference build oracle/_ref")


def product_lib():
    lib = C.CDLL(util.emu_lib_path())
    lib.dav1d_hip_lister_mask_offset.restype = C.c_long
    lib.dav1d_hip_lister_const_masks.restype = C.c_void_p
    lib.dav1d_hip_lister_block_warp.argtypes = [C.POINTER(_lib.WarpParams), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    return lib


def ref_table(name, dtype, shape):
    sz = C.c_size_t()
    p = ref.dav1d_ref_table(name.encode(), C.byref(sz))
    assert p, name
    a = np.ctypeslib.as_array((C.c_uint8 * sz.value).from_address(p)).view(dtype)
    return a.reshape(shape)


def test_geometry_tables_match_the_reference():
    lib = product_lib()
    buf = np.zeros(22 * 4 + 19 * 7 + 22 * 4 + 5 * 10 * 2, np.uint8)
    lib.dav1d_hip_lister_tables(buf.ctypes.data_as(C.c_void_p))
    bs_dim = buf[:88].reshape(22, 4)
    tx = buf[88:88 + 133].reshape(19, 7)
    max_tx = buf[221:221 + 88].reshape(22, 4)
    bsz = buf[309:].reshape(5, 10, 2)
    assert np.array_equal(bs_dim, ref_table("block_dimensions", np.uint8, (22, 4)))
    rtx = ref_table("txfm_dimensions", np.uint8, (19, 8))          # w, h, lw, lh, min, max, sub, ctx
    assert np.array_equal(tx, rtx[:, :7])
    rmax = ref_table("max_txfm_size_for_bs", np.uint8, (22, 4))
    # 4:2:2 entries of block sizes that cannot occur in 4:2:2 are 0 in the reference table (tall chroma blocks)
    legal = np.ones((22, 4), bool)
    legal[:, 2] = (rmax[:, 2] != 0) | (np.arange(22) == 21)
    assert np.array_equal(max_tx[legal], rmax[legal])
    rbsz = ref_table("block_sizes", np.uint8, (5, 10, 2))
    used = np.zeros((5, 10, 2), bool)
    for bl in range(5):
        for bp in range(10):
            if bl == 4 and bp > 3:
                continue
            if bl == 0 and bp >= 8:
                continue
            if bp == 3 and bl != 4:
                continue
            used[bl, bp, 0] = True
            used[bl, bp, 1] = 4 <= bp <= 7
    assert np.array_equal(bsz[used], rbsz[used])


def test_wedge_and_interintra_masks_match_the_reference():
    lib = product_lib()
    nb = C.c_size_t()
    blob = np.ctypeslib.as_array((C.c_uint8 * 1).from_address(lib.dav1d_hip_lister_const_masks(C.byref(nb))))
    blob = np.ctypeslib.as_array((C.c_uint8 * nb.value).from_address(blob.ctypes.data))
    ref.dav1d_ref_masks.restype = C.c_void_p
    sz = C.c_size_t()
    masks = np.ctypeslib.as_array((C.c_uint8 * 1).from_address(ref.dav1d_ref_masks(C.byref(sz))))
    masks = np.ctypeslib.as_array((C.c_uint8 * sz.value).from_address(masks.ctypes.data))
    offs = masks[:3 * 11 * 72].view(np.uint16).reshape(3, 11, 36)       # Dav1dMasks.offsets, src/wedge.h:35-38
    sizes = [(32, 32), (32, 16), (32, 8), None, (16, 32), (16, 16), (16, 8), None, (8, 32), (8, 16), (8, 8)]
    n = 0
    for c in range(3):
        sh, sv = (0, 0) if c == 0 else (1, 0) if c == 1 else (1, 1)
        for k, wh in enumerate(sizes):
            if wh is None:
                continue
            pw, ph = wh[0] >> sh, wh[1] >> sv
            for sign in range(2):
                for idx in range(16):
                    ro = int(offs[c, k, sign * 16 + idx]) * 8
                    mo = lib.dav1d_hip_lister_mask_offset(0, c, 7 + k, sign, idx)
                    assert mo >= 0 and np.array_equal(masks[ro:ro + pw * ph], blob[mo:mo + pw * ph]), ("wedge", c, wh, sign, idx)
                    n += 1
            if wh in [(32, 32), (32, 16), (16, 32), (16, 16), (16, 8), (8, 16), (8, 8)]:
                for m in range(4):
                    ro = int(offs[c, k, 32 + m]) * 8
                    mo = lib.dav1d_hip_lister_mask_offset(1, c, 7 + k, 0, m)
                    assert mo >= 0 and np.array_equal(masks[ro:ro + pw * ph], blob[mo:mo + pw * ph]), ("ii", c, wh, m)
                    n += 1
    assert n == 3 * (9 * 32 + 7 * 4)


def test_block_warp_matches_the_reference():
    lib = product_lib()
    ref.dav1d_ref_block_warp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(5)
    n_valid = 0
    for it in range(4000):
        amp = int(rng.choice([200, 2000, 12000, 32000]))
        m = rng.integers(-amp, amp + 1, size=4).astype(np.int16)
        mv = rng.integers(-4000, 4001, size=2).astype(np.int16)
        bw4, bh4 = int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([2, 4, 8, 16, 32]))
        bx, by = int(rng.integers(0, 2000)), int(rng.integers(0, 1200))
        out = np.zeros(10, np.int32)
        want_rc = ref.dav1d_ref_block_warp(m.ctypes.data, mv.ctypes.data, bw4, bh4, bx, by, out.ctypes.data)
        wm = _lib.WarpParams()
        got_rc = lib.dav1d_hip_lister_block_warp(C.byref(wm), m.ctypes.data, mv.ctypes.data, bw4, bh4, bx, by)
        assert bool(got_rc) == bool(want_rc), (it, m, got_rc, want_rc)
        assert [wm.matrix[i] for i in range(6)] == out[:6].tolist()
        if out[4 + 0] is not None and wm.matrix[2] > 0:
            assert [wm.abcd[i] for i in range(4)] == out[6:].tolist(), (it, m)
        n_valid += not want_rc
    assert n_valid > 200
