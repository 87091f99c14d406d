"""Loop restoration (Wiener) parity: batched out-of-place HIP kernel vs the reference's in-place
wiener_c driven the way lr_stripe() drives it.  Tap ranges, sizes (w 1..384, h 1..64) and all 16
edge-flag combinations follow tests/checkasm/looprestoration.c:56-135."""
import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth


def wiener_params(rng, bpc, five_tap):
    f = np.zeros((2, 8), np.int16)
    for d in range(2):
        f0 = 0 if five_tap else int(rng.integers(-5, 11))
        f1, f2 = int(rng.integers(-23, 9)), int(rng.integers(-17, 47))
        f[d, 0] = f[d, 6] = f0
        f[d, 1] = f[d, 5] = f1
        f[d, 2] = f[d, 4] = f2
    f[0, 3] = -(f[0, 0] + f[0, 1] + f[0, 2]) * 2 + (128 if bpc > 8 else 0)      # src/lr_apply_tmpl.c:55-66
    f[1, 3] = 128 - (f[1, 0] + f[1, 1] + f[1, 2]) * 2
    return f


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_wiener_matches_reference(ctx, bpc):
    oracle = util.default_oracle()
    rng = np.random.default_rng(2100 + bpc)
    W, H = 1024, 512
    src = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    lpf = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    dst = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    sp = synth.make_planes(rng, W, H, bpc, smooth=True)[0]
    lp = synth.make_planes(rng, W, H, bpc, smooth=True)[0]
    dp = synth.make_planes(rng, W, H, bpc, smooth=False)[0]
    src.upload(0, sp); lpf.upload(0, lp); dst.upload(0, dp)
    want = synth.copy_planes([dp])[0]
    stride_px = src.stride_px(0)
    tasks = []
    y = 8
    k = 0
    while y + 70 < H:
        x = 8
        while x + 400 < W:
            w = int(rng.choice([int(rng.integers(1, 385)), int(rng.integers(1, 12)), 384, 64]))
            h = int(rng.choice([int(rng.integers(1, 65)), int(rng.integers(1, 8)), 64]))
            edges = k % 16 if k < 64 else int(rng.integers(0, 16))
            five = bool(rng.integers(0, 2))
            tasks.append((x, y, w, h, 0, edges, 1 if five else 0, 0, wiener_params(rng, bpc, five)))
            x += 400
            k += 1
        y += 72
    t = np.zeros(len(tasks), api.LR_TASK)
    for i, v in enumerate(tasks):
        t[i] = v
    bps = want.itemsize
    for i in range(len(t)):
        x, y, w, h = (int(t[i][k]) for k in ("x", "y", "w", "h"))
        work = synth.copy_planes([sp])[0]                                     # the reference filters in place, inside the picture
        left = np.ascontiguousarray(sp[y:y + h, x - 4:x])
        L = np.zeros((8, stride_px), sp.dtype)
        L[0], L[1] = lp.base[y - 2] if lp.base is not None else lp[y - 2], lp.base[y - 1] if lp.base is not None else lp[y - 1]
        L[6], L[7] = (lp.base if lp.base is not None else lp)[y + h], (lp.base if lp.base is not None else lp)[y + h + 1]
        filt = np.ascontiguousarray(t[i]["filter"])
        oracle.call(bpc, "wiener", int(t[i]["type"]), 0, work.ctypes.data + (y * stride_px + x) * bps, work.strides[0],
                    left, L.ctypes.data + x * bps, w, h, filt, int(t[i]["edges"]))
        want[y:y + h, x:x + w] = work[y:y + h, x:x + w]
    ctx.lr_batch(dst, src, lpf, t)
    got = dst.download(0)
    bad = np.argwhere(got != want)
    if len(bad):
        yy, xx = bad[0]
        hit = [tuple(t[i])[:8] for i in range(len(t)) if t[i]["x"] <= xx < t[i]["x"] + t[i]["w"] and t[i]["y"] <= yy < t[i]["y"] + t[i]["h"]]
        raise AssertionError("mismatch at (%d,%d): got %d want %d (%d px) task %s" % (xx, yy, got[yy, xx], want[yy, xx], len(bad), hit[:1]))
    for o in (src, lpf, dst):
        o.free()


def _sgr_task(rng, sgr_params, x, y, w, h, edges):
    set_idx = int(rng.integers(0, 16))
    s0, s1 = int(sgr_params[set_idx][0]), int(sgr_params[set_idx][1])
    typ = 2 + (1 if not s0 else 0 if not s1 else 2)          # 2: 5x5, 3: 3x3, 4: mix
    w0 = int(rng.integers(-96, 32))
    w1 = 128 - (w0 + int(rng.integers(-32, 96)))
    f = np.zeros((2, 8), np.int16)
    f[0, :4] = (s0, s1, w0, w1)
    return (x, y, w, h, 0, edges, typ, 0, f)


def _check_sgr(ctx, bpc, rng, make_tasks):
    import struct
    oracle = util.default_oracle()
    import golden_cases
    sgr_params = np.array(golden_cases.SGR_PARAMS, np.uint16)      # == av1_sgr_params, checked against the reference in test_abi
    W, H = 1024, 512
    src = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    lpf = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    dst = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    sp = synth.make_planes(rng, W, H, bpc, smooth=True)[0]
    lp = synth.make_planes(rng, W, H, bpc, smooth=True)[0]
    dp = synth.make_planes(rng, W, H, bpc, smooth=False)[0]
    # some flat and some noisy areas so that the whole x_by_x range is exercised
    sp[:, 300:500] = rng.integers(0, 1 << bpc, size=(sp.shape[0], 200))
    sp[100:200, :] = sp[100, 0]
    src.upload(0, sp); lpf.upload(0, lp); dst.upload(0, dp)
    want = synth.copy_planes([dp])[0]
    stride_px = src.stride_px(0)
    rows = make_tasks(rng, sgr_params, W, H)
    t = np.zeros(len(rows), api.LR_TASK)
    for i, v in enumerate(rows):
        t[i] = v
    bps = want.itemsize
    lbase = lp.base
    for i in range(len(t)):
        x, y, w, h = (int(t[i][n]) for n in ("x", "y", "w", "h"))
        work = synth.copy_planes([sp])[0]
        left = np.ascontiguousarray(sp[y:y + h, x - 4:x])
        L = np.zeros((8, stride_px), sp.dtype)
        L[0], L[1], L[6], L[7] = lbase[y - 2], lbase[y - 1], lbase[y + h], lbase[y + h + 1]
        s0, s1, w0, w1 = (int(v) for v in t[i]["filter"][0][:4])
        params = np.frombuffer(struct.pack("<IIhh", s0, s1, w0, w1) + b"\0" * 20, np.uint8).copy()
        oracle.call(bpc, "sgr", int(t[i]["type"]) - 2, 0, work.ctypes.data + (y * stride_px + x) * bps, work.strides[0],
                    left, L.ctypes.data + x * bps, w, h, params, int(t[i]["edges"]))
        want[y:y + h, x:x + w] = work[y:y + h, x:x + w]
    ctx.lr_batch(dst, src, lpf, t)
    got = dst.download(0)
    bad = np.argwhere(got != want)
    if len(bad):
        yy, xx = bad[0]
        hit = [tuple(t[i])[:8] + (tuple(t[i]["filter"][0][:4]),) for i in range(len(t))
               if t[i]["x"] <= xx < t[i]["x"] + t[i]["w"] and t[i]["y"] <= yy < t[i]["y"] + t[i]["h"]]
        raise AssertionError("mismatch at (%d,%d): got %d want %d (%d px) task %s" % (xx, yy, got[yy, xx], want[yy, xx], len(bad), hit[:1]))
    assert set(t["type"]) == {2, 3, 4}
    for o in (src, lpf, dst):
        o.free()


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_sgr_matches_reference(ctx, bpc):
    """sgr_5x5 / sgr_3x3 / sgr_mix with the parameter sets of dav1d_sgr_params and the weight ranges of
    tests/checkasm/looprestoration.c:137-195."""
    def make(rng, sgr_params, W, H):
        out = []
        y = 8
        while y + 70 < H:
            x = 8
            while x + 400 < W:
                w = int(rng.choice([int(rng.integers(1, 385)), int(rng.integers(1, 12)), 384, 64]))
                h = int(rng.choice([int(rng.integers(1, 65)), int(rng.integers(1, 9)), 64]))
                k = len(out)
                out.append(_sgr_task(rng, sgr_params, x, y, w, h, k % 16 if k < 32 else int(rng.integers(0, 16))))
                x += 400
            y += 72
        return out
    _check_sgr(ctx, bpc, np.random.default_rng(2500 + bpc), make)


@pytest.mark.parametrize("bpc", [8, 10])
def test_sgr_units_of_a_row_share_waves(ctx, bpc):
    """Rows of units with the same y and height — adjacent, with gaps, one to seventy pixels wide, of all three filter types and with
    every edge combination side by side: the kernel lays the units of a row out in one run of columns and cuts the run into waves
    (lr.hip), so a wave holds the end of one unit, whole small units and the start of the next."""
    def make(rng, sgr_params, W, H):
        out = []
        y = 8
        while y + 70 < H:
            h = int(rng.choice([64, 64, 56, int(rng.integers(1, 65)), 2, 5]))
            x = 8
            while True:
                w = int(rng.choice([64, 64, 32, int(rng.integers(1, 8)), int(rng.integers(1, 71)), 1, 60, 61, 62, 63]))
                if x + w + 8 > W:
                    break
                out.append(_sgr_task(rng, sgr_params, x, y, w, h, int(rng.integers(0, 16))))
                x += w + int(rng.choice([0, 0, 0, 3, 17]))
            y += 72
        return out
    _check_sgr(ctx, bpc, np.random.default_rng(2600 + bpc), make)
