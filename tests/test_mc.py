"""mc / mct / compound parity: HIP kernels (through the C ABI) vs the reference C functions.

The oracle side replays what the reference driver mc() does (src/recon_tmpl.c:938-989):
emu_edge into a 192-pixel-stride scratch when the filter window leaves the visible plane,
then dsp->mc.mc[filter] / mct[filter].  Shapes, phases and bit depths follow
tests/checkasm/mc.c:58-122."""
import numpy as np
import pytest

import util
from dav1d_amd import api

SIZES_PUT = [2, 4, 8, 16, 32, 64, 128]


def _oracle_mc(oracle, bpc, refplane, vis_w, vis_h, task, dst_block=None, tmp=None):
    """One mc() of the reference driver on a host plane (2-D array, padded)."""
    pd = refplane.dtype
    bps = pd.itemsize
    w, h, mx, my, f = (int(task[k]) for k in ("w", "h", "mx", "my", "filter_2d"))
    dx, dy = int(task["src_x"]), int(task["src_y"])
    if dx < (3 if mx else 0) or dy < (3 if my else 0) or dx + w + (4 if mx else 0) > vis_w or dy + h + (4 if my else 0) > vis_h:
        emu = np.zeros((192 + 8, 192), pd)
        oracle.call(bpc, "emu_edge", 0, 0, w + (7 if mx else 0), h + (7 if my else 0), vis_w, vis_h,
                    dx - (3 if mx else 0), dy - (3 if my else 0), emu.ctypes.data, 192 * bps,
                    refplane.ctypes.data, refplane.strides[0])
        src_ptr = emu.ctypes.data + ((192 * 3 if my else 0) + (3 if mx else 0)) * bps
        src_stride = 192 * bps
    else:
        src_ptr = refplane.ctypes.data + dy * refplane.strides[0] + dx * bps
        src_stride = refplane.strides[0]
    if dst_block is not None:
        oracle.call(bpc, "mc", f, 0, dst_block.ctypes.data, dst_block.strides[0], src_ptr, src_stride, w, h, mx, my)
    else:
        oracle.call(bpc, "mct", f, 0, tmp.ctypes.data, src_ptr, src_stride, w, h, mx, my)


def _gen_tasks(rng, n, vis_w, vis_h, DW, DH, kind, sizes):
    """Random tasks; PUT blocks packed without overlap into a DW x DH dst plane."""
    t = np.zeros(n, api.MC_TASK)
    pos = []
    x = y = row_h = 0
    prep_off = 0
    k = 0
    for i in range(n):
        w = int(rng.choice(sizes, p=[.12, .2, .22, .2, .14, .08, .04]))
        h = int(rng.choice([v for v in SIZES_PUT if max(w // 4, 2) <= v <= min(w * 4, 128)]))
        if kind == 1:
            w = max(w, 4)
        if x + w > DW:
            x = 0; y += row_h; row_h = 0
        if y + h > DH:
            break
        mode = rng.integers(0, 4)
        mx = 0 if mode in (0, 2) else int(rng.integers(1, 16))
        my = 0 if mode in (0, 1) else int(rng.integers(1, 16))
        if rng.integers(0, 4) == 0:   # exercise edge emulation on every side
            sx = int(rng.choice([-w - 5, -3, -1, vis_w - w + 1, vis_w - 2, vis_w + 9, int(rng.integers(0, vis_w))]))
            sy = int(rng.choice([-h - 5, -2, vis_h - h + 2, vis_h - 1, vis_h + 20, int(rng.integers(0, vis_h))]))
        else:
            sx = int(rng.integers(4, max(5, vis_w - w - 4)))
            sy = int(rng.integers(4, max(5, vis_h - h - 4)))
        t[k]["src_x"], t[k]["src_y"], t[k]["w"], t[k]["h"] = sx, sy, w, h
        t[k]["mx"], t[k]["my"], t[k]["filter_2d"] = mx, my, int(rng.integers(0, 10))
        t[k]["kind"], t[k]["plane"], t[k]["ref"] = kind, 0, 0
        if kind == 0:
            pos.append((x, y))
        else:
            pos.append(prep_off)
            prep_off += w * h
        x += w
        row_h = max(row_h, h)
        k += 1
    return t[:k], pos, prep_off


@pytest.mark.parametrize("twin", [False, True], ids=["raster", "tiled"])
@pytest.mark.parametrize("bpc", [8, 10, 12])
@pytest.mark.parametrize("kind", [0, 1], ids=["put", "prep"])
def test_mc_matches_reference(ctx, bpc, kind, twin):
    """twin: the reference picture is read through its tiled twin (8x8 tiles, dav1d_hip_picture_retile) — the same pixels must
    come out, windows that leave the picture on any side included."""
    oracle = util.default_oracle()
    rng = np.random.default_rng(77 + bpc * 2 + kind)
    vis_w, vis_h = 200, 150                      # visible size; allocation is padded to 256 x 256
    DW = DH = 512
    pd = util.pix_dtype(bpc)
    ref = ctx.picture(vis_w, vis_h, api.LAYOUT_I400, bpc)
    refplane = rng.integers(0, 1 << bpc, size=ref.padded_shape(0)).astype(pd)   # padding is NOT edge-replicated
    ref.upload(0, refplane)
    if twin:
        ref.retile()
        assert ref.pic.twin_ok and ref.pic.twin[0]
    dst = ctx.picture(DW, DH, api.LAYOUT_I400, bpc)
    dplane = rng.integers(0, 1 << bpc, size=dst.padded_shape(0)).astype(pd)
    dst.upload(0, dplane)
    n = 60 if ctx.backend == "emu" else 400
    tasks, pos, prep_sz = _gen_tasks(rng, n, vis_w, vis_h, DW, DH, kind, SIZES_PUT)
    assert len(tasks) > 20
    want_plane = dplane.copy()
    want_prep = np.zeros(max(prep_sz, 1), np.int16)
    sp = dst.stride_px(0)
    for i, t in enumerate(tasks):
        if kind == 0:
            x, y = pos[i]
            tasks[i]["dst_off"] = y * sp + x
            _oracle_mc(oracle, bpc, refplane, vis_w, vis_h, t, dst_block=want_plane[y:, x:])
        else:
            tasks[i]["dst_off"] = pos[i]
            _oracle_mc(oracle, bpc, refplane, vis_w, vis_h, t, tmp=want_prep[pos[i]:])
    prep = ctx.buffer(max(prep_sz, 8) * 2)
    prep.zero()
    ctx.mc_batch(dst, [ref], tasks, prep)
    got_plane = dst.download(0)
    got_prep = prep.download(np.int16, max(prep_sz, 1))
    if kind == 0:
        bad = np.argwhere(got_plane != want_plane)
        if len(bad):
            yy, xx = bad[0]
            hit = [(tuple(tasks[i]), pos[i]) for i in range(len(tasks))
                   if pos[i][0] <= xx < pos[i][0] + tasks[i]["w"] and pos[i][1] <= yy < pos[i][1] + tasks[i]["h"]]
            raise AssertionError("put mismatch at (%d,%d): got %d want %d task %s" %
                                 (xx, yy, got_plane[yy, xx], want_plane[yy, xx], hit[:1]))
    else:
        bad = np.flatnonzero(got_prep != want_prep)
        if len(bad):
            b = bad[0]
            hit = [tuple(tasks[i]) for i in range(len(tasks)) if pos[i] <= b < pos[i] + int(tasks[i]["w"]) * int(tasks[i]["h"])]
            raise AssertionError("prep mismatch at %d: got %d want %d task %s" % (b, got_prep[b], want_prep[b], hit[:1]))
        assert np.array_equal(got_plane, dplane), "prep must not touch the dst picture"
    for o in (ref, dst, prep):
        o.free()


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_compound_matches_reference(ctx, bpc):
    """avg / w_avg / mask / w_mask{444,422,420} (tests/checkasm/mc.c:289-480)."""
    oracle = util.default_oracle()
    rng = np.random.default_rng(5 + bpc)
    pd = util.pix_dtype(bpc)
    DW = DH = 256
    dst = ctx.picture(DW, DH, api.LAYOUT_I400, bpc)
    dplane = rng.integers(0, 1 << bpc, size=dst.padded_shape(0)).astype(pd)
    dst.upload(0, dplane)
    want = dplane.copy()
    sp = dst.stride_px(0)
    bias = 8192 if bpc > 8 else 0
    ib = 4 if bpc == 8 else 14 - bpc
    tasks = np.zeros(64, api.COMP_TASK)
    preps, masks = [], []
    prep_off = mask_off = 0
    x = y = row_h = 0
    k = 0
    want_mask = []
    for kind, ss in [(0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 2)] * 6:
        w = int(rng.choice([4, 8, 16, 32, 64, 128]))
        h = int(rng.choice([v for v in [4, 8, 16, 32, 64, 128] if w // 4 <= v <= w * 4]))
        if x + w > DW:
            x = 0; y += row_h; row_h = 0
        if y + h > DH or k >= len(tasks):
            break
        # the value range prep produces (reference src/mc_tmpl.c:41-48)
        t1 = (rng.integers(0, 1 << bpc, size=w * h) << ib).astype(np.int32) - bias
        t2 = (rng.integers(0, 1 << bpc, size=w * h) << ib).astype(np.int32) - bias
        t1 = t1.astype(np.int16); t2 = t2.astype(np.int16)
        arg = 0
        m_in = np.zeros(w * h, np.uint8)
        m_out = np.zeros(w * h, np.uint8)
        blk = want[y:, x:]
        if kind == 0:
            oracle.call(bpc, "avg", 0, 0, blk.ctypes.data, want.strides[0], t1, t2, w, h)
        elif kind == 1:
            arg = int(rng.integers(1, 16))
            oracle.call(bpc, "w_avg", 0, 0, blk.ctypes.data, want.strides[0], t1, t2, w, h, arg)
        elif kind == 2:
            m_in = rng.integers(0, 65, size=w * h).astype(np.uint8)
            oracle.call(bpc, "mask", 0, 0, blk.ctypes.data, want.strides[0], t1, t2, w, h, m_in)
        else:
            arg = int(rng.integers(0, 2))
            oracle.call(bpc, "w_mask", ss, 0, blk.ctypes.data, want.strides[0], t1, t2, w, h, m_out, arg)
        tasks[k] = (y * sp + x, prep_off, prep_off + w * h, mask_off, w, h, kind, 0, arg, ss, 0)
        preps += [t1, t2]
        masks.append(m_in)
        want_mask.append(m_out if kind == 3 else m_in)
        prep_off += 2 * w * h
        mask_off += w * h
        x += w
        row_h = max(row_h, h)
        k += 1
    tasks = tasks[:k]
    dprep = ctx.buffer_from(np.concatenate(preps))
    dmask = ctx.buffer_from(np.concatenate(masks))
    ctx.comp_batch(dst, tasks, dprep, dmask)
    got = dst.download(0)
    got_mask = dmask.download(np.uint8, mask_off)
    assert np.array_equal(got, want)
    # only the sub-sampled part of a w_mask output is defined
    off = 0
    for i, t in enumerate(tasks):
        w, h = int(t["w"]), int(t["h"])
        n = w * h
        if t["kind"] == 3:
            n = (w >> (1 if t["ss"] else 0)) * (h >> (1 if t["ss"] == 2 else 0))
        assert np.array_equal(got_mask[off:off + n], want_mask[i][:n]), ("mask", i, tuple(t))
        off += w * h
    for o in (dst, dprep, dmask):
        o.free()
