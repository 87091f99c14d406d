"""Deblocking parity: batched HIP kernel (one task = one dsp->lf.loop_filter_sb call) vs the
reference functions, run in the reference's order (all column edges, then all row edges).
Masks / levels / limit LUT follow tests/checkasm/loopfilter.c:96-160."""
import ctypes as C

import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth


def make_lut(sharp):
    e, i = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
    for level in range(64):
        limit = level
        if sharp > 0:
            limit >>= (sharp + 3) >> 2
            limit = min(limit, 9 - sharp)
        limit = max(limit, 1)
        i[level] = limit
        e[level] = 2 * (level + 2) + limit
    return e, i


class LutStruct(C.Structure):       # Av1FilterLUT, reference src/lf_mask.h:36-40
    _fields_ = [("e", C.c_uint8 * 64), ("i", C.c_uint8 * 64), ("sharp", C.c_uint64 * 2)]


def structured_plane(rng, shape, bpc):
    """Piecewise-flat content with per-block noise levels so that flat16 / flat8 / hev / normal paths all fire."""
    h, w = shape
    bd8 = bpc - 8
    base = rng.integers(0, 1 << bpc, size=((h + 15) // 16, (w + 15) // 16))
    amp = rng.choice([0, 1 << bd8, 3 << bd8, 12 << bd8, 1 << bpc], size=base.shape, p=[.2, .25, .2, .2, .15])
    b = np.kron(base, np.ones((16, 16), np.int64))[:h, :w]
    a = np.kron(amp, np.ones((16, 16), np.int64))[:h, :w]
    noise = (rng.random((h, w)) * (a + 1)).astype(np.int64) - a // 2
    return np.clip(b + noise, 0, (1 << bpc) - 1)


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_loop_filter_sb_matches_reference(ctx, bpc):
    oracle = util.default_oracle()
    rng = np.random.default_rng(800 + bpc)
    w, h = 256, 256
    pic = ctx.picture(w, h, api.LAYOUT_I420, bpc)
    planes = synth.make_planes(rng, w, h, bpc, smooth=False)
    for pl in range(3):
        planes[pl][:, :] = structured_plane(rng, planes[pl].shape, bpc)
        pic.upload(pl, planes[pl])
    b4_stride = 96
    lvl = rng.integers(0, 64, size=(80, b4_stride, 4)).astype(np.uint8)
    lvl[rng.random(lvl.shape) < 0.15] = 0
    e, i = make_lut(int(rng.integers(0, 8)))
    lut = LutStruct()
    lut.e[:] = list(e)
    lut.i[:] = list(i)
    tasks = []
    for pl in range(3):
        ph, pw = (h, w) if pl == 0 else (h // 2, w // 2)
        sp = pic.stride_px(pl)
        nwords = 3 if pl == 0 else 2
        comp = {0: (0, 1), 1: (2, 2), 2: (3, 3)}[pl]
        for d in (0, 1):
            length = (ph if d == 0 else pw) // 4
            nunits = min(32, length)
            for pos in range(16, (pw if d == 0 else ph) - 8, 16):     # edges 16 pixels apart: no interaction inside a pass
                for seg in range(0, length, 32):
                    vm = [0, 0, 0]
                    for u in range(min(nunits, length - seg)):
                        idx = int(rng.integers(0, nwords + 1))
                        if idx:
                            vm[idx - 1] |= 1 << u
                    if d == 0:
                        x, y = pos, seg * 4
                    else:
                        x, y = seg * 4, pos
                    tasks.append((y * sp + x, (y // 4) * b4_stride + x // 4, vm, pl, d, comp[d], 0))
    t = np.zeros(len(tasks), api.LF_TASK)
    for k, v in enumerate(tasks):
        t[k] = v
    # oracle: every column-edge call, then every row-edge call, in place on host copies
    want = synth.copy_planes(planes)
    bps = want[0].itemsize
    for d in (0, 1):
        for k in range(len(t)):
            if t[k]["dir"] != d:
                continue
            pl = int(t[k]["plane"])
            vm = (C.c_uint32 * 4)(*[int(v) for v in t[k]["vmask"]], 0)
            lp = lvl.ctypes.data + int(t[k]["lvl_off"]) * 4 + int(t[k]["lvl_comp"])
            oracle.call(bpc, "loop_filter_sb", 1 if pl else 0, d, want[pl].ctypes.data + int(t[k]["dst_off"]) * bps,
                        want[pl].strides[0], vm, lp, b4_stride, C.byref(lut), 32)
    dlvl = ctx.buffer_from(lvl)
    ctx.lf_batch(pic, t[rng.permutation(len(t))], dlvl, b4_stride, e, i)
    changed = 0
    for pl in range(3):
        got = pic.download(pl)
        bad = np.argwhere(got != want[pl])
        assert not len(bad), "plane %d differs at %s: got %d want %d" % (pl, bad[0], got[tuple(bad[0])], want[pl][tuple(bad[0])])
        changed += int((want[pl] != planes[pl]).sum())
    assert changed > 1000, "the case must actually filter something"
    pic.free(); dlvl.free()
