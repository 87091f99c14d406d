"""splat_mv and save_tmvs of the reference's Dav1dRefmvsDSPContext (src/refmvs.c:763-803, 914-923) on a frame-level map in
device memory (csrc/refmvs.hip) against the reference's own C functions driven with its row-pointer arrays."""
import ctypes as C

import numpy as np
import pytest

import util

pytestmark = pytest.mark.skipif(util.ref_lib() is None, reason="needs the reference build oracle/_ref")

SPLAT = np.dtype([("bx4", "<u2"), ("by4", "<u2"), ("bw4", "u1"), ("bh4", "u1"), ("pad", "u1", (2,)), ("rmv", "<u4", (3,))])
assert SPLAT.itemsize == 20
BLOCK_W4 = [32, 32, 16, 16, 16, 16, 8, 8, 8, 8, 4, 4, 4, 4, 4, 2, 2, 2, 2, 1, 1, 1]      # dav1d_block_dimensions[bs][0] / [1] (== h_bs_dim, pinned in test_host_tables.py)
BLOCK_H4 = [32, 16, 32, 16, 8, 4, 16, 8, 4, 2, 16, 8, 4, 2, 1, 8, 4, 2, 1, 4, 2, 1]


# (4x4 columns, rows, row stride in 4x4 units, seed, ref_sign): a 384 x 256 frame (three 128-wide, two 128-high superblocks); a 1920 x 1088 one;
# one whose height leaves a partial superblock row of unit rows; every sign / no sign at all
CASES = [(96, 64, 128, 5, [1, 0, 1, 1, 0, 1, 0]), (480, 272, 512, 6, [0, 1, 1, 0, 1, 0, 1]), (160, 88, 192, 7, [1, 1, 1, 1, 1, 1, 1]),
         (64, 40, 64, 8, [0, 0, 0, 0, 0, 0, 0]), (328, 184, 352, 9, [1, 0, 0, 1, 0, 0, 1])]


@pytest.mark.parametrize("w4,h4,stride4,seed,signs", CASES, ids=["384x256", "1920x1088", "640x352", "256x160-no-signs", "1312x736"])
def test_splat_and_save_tmvs_match_the_reference(ctx, w4, h4, stride4, seed, signs):
    ref = util.ref_lib()
    if ctx.backend == "emu" and w4 * h4 > 20000:
        pytest.skip("a minute of emulation and of Python row-pointer calls: GPU only")
    rng = np.random.default_rng(seed)
    host = np.zeros((h4, stride4, 3), np.uint32)            # the reference's rows (12-byte records)
    # ---- blocks: a grid of 32x32-pixel cells, each cut into random legal block sizes; one refmvs_block per block
    tasks = []
    for gy in range(0, h4, 8):
        for gx in range(0, w4, 8):
            bs = int(rng.choice([7, 12, 17, 21, 13, 16, 8, 11]))           # 32x32 .. 4x4 and some rectangles
            bw, bh = min(BLOCK_W4[bs], 8), min(BLOCK_H4[bs], 8)
            for y in range(gy, gy + 8, bh):
                for x in range(gx, gx + 8, bw):
                    mvs = rng.integers(-6000, 6000, size=4).astype(np.int16)
                    if rng.integers(0, 4) == 0:
                        mvs[:] = rng.integers(-300, 300, size=4)
                    refs = rng.integers(-1, 8, size=2).astype(np.int8)
                    rec = np.zeros(12, np.uint8)
                    rec[0:8] = mvs.view(np.uint8)
                    rec[8:10] = refs.view(np.uint8)
                    rec[10], rec[11] = bs, int(rng.integers(0, 3))
                    t = np.zeros((), SPLAT)
                    t["bx4"], t["by4"], t["bw4"], t["bh4"] = x, y, bw, bh
                    t["rmv"] = rec.view(np.uint32)
                    tasks.append(t)
    tasks = np.array(tasks, SPLAT)
    # ---- reference splat_mv: row pointers into `host`
    ref.dav1d_ref_refmvs_splat.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    for t in tasks:
        rows = (C.c_void_p * int(t["bh4"]))(*[host[int(t["by4"]) + k].ctypes.data for k in range(int(t["bh4"]))])
        rec = np.ascontiguousarray(t["rmv"])
        ref.dav1d_ref_refmvs_splat(rows, rec.ctypes.data, int(t["bx4"]), int(t["bw4"]), int(t["bh4"]))
    dev = ctx.buffer(host.nbytes)
    dev.zero()
    assert ctx.lib.dav1d_hip_refmvs_splat_batch(ctx.h, dev.ptr, stride4, tasks.ctypes.data, len(tasks)) == 0
    got = dev.download(np.uint32, host.size).reshape(host.shape)
    assert np.array_equal(got, host), "splat_mv"
    # ---- save_tmvs: the reference per superblock row of 16 unit rows, rr[(y & 15) * 2] = the 4x4 row 2 * y
    iw8, ih8 = w4 // 2, h4 // 2
    rp_stride = ((w4 * 4 + 127) & ~127) >> 3
    sign = np.array(signs, np.uint8)
    want = np.zeros((ih8, rp_stride, 5), np.uint8)
    ref.dav1d_ref_refmvs_save_tmvs.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    for y0 in range(0, ih8, 16):
        rr = (C.c_void_p * 32)(*[host[min(2 * y0 + k, h4 - 1)].ctypes.data for k in range(32)])
        # two tile columns: [0, 24) and [24, iw8)
        for c0, c1 in ((0, min(24, iw8)), (min(24, iw8), iw8)):
            if c1 <= c0:
                continue
            ref.dav1d_ref_refmvs_save_tmvs(want[y0].ctypes.data, rp_stride, rr, sign.ctypes.data, c1, min(y0 + 16, ih8), c0, y0)
    rp = ctx.buffer(want.nbytes)
    rp.zero()
    assert ctx.lib.dav1d_hip_refmvs_save_tmvs(ctx.h, rp.ptr, rp_stride, dev.ptr, stride4, sign.ctypes.data, 0, iw8, 0, ih8) == 0
    got_rp = rp.download(np.uint8, want.size).reshape(want.shape)
    assert np.array_equal(got_rp[:, :iw8], want[:, :iw8]), "save_tmvs"
    assert (want[:, :iw8, 4] == 0).any() and (want[:, :iw8, 4].any() or not any(signs))
    dev.free()
    rp.free()

