"""Intra prediction parity: batched HIP kernel (task = prepare_intra_edges + intra_pred / cfl / pal) vs
the reference's own dav1d_prepare_intra_edges + DSP entries.  Modes, sizes, angles and flags follow
tests/checkasm/ipred.c:78-296 (all 14 predictors x w,h in 4..64, 27 angles x edge-filter / smooth bits)."""
import ctypes as C

import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth

CELL = 192


def _prepare(lib, bpc):
    f = getattr(lib, "dav1d_prepare_intra_edges_%dbpc" % (8 if bpc == 8 else 16))
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 6 + [C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                  C.c_int, C.c_int, C.c_int, C.c_void_p] + ([C.c_int] if bpc > 8 else [])
    return f


def oracle_task(oracle, bpc, planes, t, layout, pal_idx):
    pl = int(t["plane"])
    p = planes[pl]
    bps = p.itemsize
    dptr = p.ctypes.data + int(t["dst_off"]) * bps
    w, h = int(t["tw"]) * 4, int(t["th"]) * 4
    kind = int(t["kind"])
    if kind == 2:
        pal = np.array(t["pal"], p.dtype)
        idx = pal_idx[int(t["aux_off"]):]
        oracle.call(bpc, "pal_pred", 0, 0, dptr, p.strides[0], pal, idx.ctypes.data, w, h)
        return
    edge = np.zeros(1024, p.dtype)
    tl = edge.ctypes.data + 512 * bps
    fl = int(t["flags"])
    edge_flags = (1 if fl & 4 else 0) | (8 if fl & 8 else 0)
    angle = C.c_int(int(t["angle"]) if kind == 0 else 0)
    mode = int(t["mode"]) if kind == 0 else 0
    args = [int(t["x4"]), fl & 1, int(t["y4"]), (fl >> 1) & 1, int(t["w4"]), int(t["h4"]), edge_flags, dptr, p.strides[0], None,
            mode, C.byref(angle), int(t["tw"]), int(t["th"]), (fl >> 4) & 1, tl]
    if bpc > 8:
        args.append((1 << bpc) - 1)
    m = _prepare(oracle.lib, bpc)(*args)
    if kind == 0:
        a = angle.value | (512 if fl & 32 else 0) | (1024 if fl & 16 else 0)
        oracle.call(bpc, "intra_pred", m, 0, dptr, p.strides[0], tl, w, h, a, int(t["max_w"]), int(t["max_h"]))
    else:
        ac = np.zeros(32 * 32, np.int16)
        yp = planes[0]
        oracle.call(bpc, "cfl_ac", layout - 1, 0, ac, yp.ctypes.data + int(t["aux_off"]) * yp.itemsize, yp.strides[0],
                    int(t["max_w"]), int(t["max_h"]), w, h)
        oracle.call(bpc, "cfl_pred", m, 0, dptr, p.strides[0], tl, w, h, ac, int(t["angle"]))


def gen_batches(rng, pic, bpc, n_batches):
    """Independent tasks: one block per 192x192 cell of plane 0 (blocks sit at (64, 64) inside their cell)."""
    W = pic.w
    cells = [(cx, cy) for cy in range(0, pic.h - CELL + 1, CELL) for cx in range(0, W - CELL + 1, CELL)]
    sp = pic.stride_px(0)
    sizes = [4, 8, 16, 32, 64]
    for b in range(n_batches):
        t = np.zeros(len(cells), api.IPRED_TASK)
        for k, (cx, cy) in enumerate(cells):
            x, y = cx + 64, cy + 64
            w = int(rng.choice(sizes))
            h = int(rng.choice([v for v in sizes if w // 4 <= v <= w * 4]))
            mode = int(rng.integers(0, 14))
            if mode == 13 and (w > 32 or h > 32):
                mode = 12
            flags = int(rng.integers(0, 64))
            if rng.integers(0, 3):
                flags |= 3                                    # mostly both neighbours available
            t[k]["dst_off"] = y * sp + x
            t[k]["x4"], t[k]["y4"] = x // 4, y // 4
            # tile end: usually far, sometimes clipping the top-right / bottom-left / edge extension
            t[k]["w4"] = x // 4 + int(rng.choice([w // 4, w // 4 + 1, 2 * (w // 4), 40]))
            t[k]["h4"] = y // 4 + int(rng.choice([h // 4, h // 4 + 1, 2 * (h // 4), 40]))
            t[k]["tw"], t[k]["th"], t[k]["mode"] = w // 4, h // 4, mode
            t[k]["angle"] = int(rng.integers(-3, 4)) if 1 <= mode <= 8 else (int(rng.integers(0, 5)) if mode == 13 else 0)
            t[k]["flags"], t[k]["plane"], t[k]["kind"] = flags, 0, 0
            t[k]["max_w"] = int(rng.choice([w, max(4, w // 2), 4 * 40]))
            t[k]["max_h"] = int(rng.choice([h, max(4, h // 2), 4 * 40]))
        yield t


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_intra_pred_matches_reference(ctx, bpc):
    oracle = util.default_oracle()
    rng = np.random.default_rng(1500 + bpc)
    W = H = 768
    pic = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    plane = synth.make_planes(rng, W, H, bpc, smooth=True)[0]
    pic.upload(0, plane)
    want = synth.copy_planes([plane])
    seen = set()
    for t in gen_batches(rng, pic, bpc, 24 if ctx.backend == "emu" else 120):
        for k in range(len(t)):
            oracle_task(oracle, bpc, want, t[k], 0, None)
            seen.add(int(t[k]["mode"]))
        ctx.ipred_batch(pic, t)
        got = pic.download(0)
        bad = np.argwhere(got != want[0])
        if len(bad):
            yy, xx = bad[0]
            k = (yy // CELL) * (W // CELL) + xx // CELL
            raise AssertionError("mismatch at (%d,%d): got %d want %d; task %s" % (xx, yy, got[yy, xx], want[0][yy, xx], tuple(t[k])))
    assert len(seen) == 14
    pic.free()


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_cfl_and_palette_match_reference(ctx, bpc):
    oracle = util.default_oracle()
    rng = np.random.default_rng(1700 + bpc)
    W = H = 768
    layout = api.LAYOUT_I420
    pic = ctx.picture(W, H, layout, bpc)
    planes = synth.make_planes(rng, W, H, bpc, smooth=True)
    for pl in range(3):
        pic.upload(pl, planes[pl])
    want = synth.copy_planes(planes)
    pal_idx = (rng.integers(0, 8, size=1 << 16) | (rng.integers(0, 8, size=1 << 16) << 4)).astype(np.uint8)
    dpal = ctx.buffer_from(pal_idx)
    spc, spy = pic.stride_px(1), pic.stride_px(0)
    cells = [(cx, cy) for cy in range(0, H // 2 - CELL + 1, CELL) for cx in range(0, W // 2 - CELL + 1, CELL)]
    for b in range(12 if ctx.backend == "emu" else 60):
        t = np.zeros(len(cells), api.IPRED_TASK)
        off = 0
        for k, (cx, cy) in enumerate(cells):
            x, y = cx + 64, cy + 64                       # chroma position
            w = int(rng.choice([4, 8, 16, 32]))
            h = int(rng.choice([v for v in [4, 8, 16, 32] if w // 4 <= v <= w * 4]))
            pl = int(rng.integers(1, 3))
            t[k]["dst_off"] = y * spc + x
            t[k]["tw"], t[k]["th"], t[k]["plane"] = w // 4, h // 4, pl
            if rng.integers(0, 3) == 0:
                t[k]["kind"] = 2
                t[k]["aux_off"] = off
                off += w * h // 2 + 16
                t[k]["pal"] = rng.integers(0, 1 << bpc, size=8)
            else:
                t[k]["kind"] = 1
                t[k]["aux_off"] = 2 * y * spy + 2 * x
                t[k]["x4"], t[k]["y4"], t[k]["w4"], t[k]["h4"] = x // 4, y // 4, x // 4 + 40, y // 4 + 40
                t[k]["flags"] = int(rng.choice([0, 1, 2, 3, 3, 3]))
                t[k]["angle"] = int(rng.integers(-16, 17))
                t[k]["max_w"] = int(rng.integers(0, w // 4))        # w_pad
                t[k]["max_h"] = int(rng.integers(0, h // 4))        # h_pad
        for k in range(len(t)):
            oracle_task(oracle, bpc, want, t[k], layout, pal_idx)
        ctx.ipred_batch(pic, t, dpal)
        for pl in (1, 2):
            got = pic.download(pl)
            bad = np.argwhere(got != want[pl])
            if len(bad):
                yy, xx = bad[0]
                k = (yy // CELL) * ((W // 2) // CELL) + xx // CELL
                raise AssertionError("plane %d mismatch at (%d,%d): got %d want %d; task %s" %
                                     (pl, xx, yy, got[yy, xx], want[pl][yy, xx], tuple(t[k])))
    pic.free(); dpal.free()
