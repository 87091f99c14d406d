"""The remaining Dav1dMCDSPContext entries: blend / blend_v / blend_h (+ the PUT_TMP predictions that feed
them), warp8x8 / warp8x8t, mc_scaled / mct_scaled, resize and emu_edge -- HIP kernels through the C ABI vs the
reference C functions, driven the way the reference drivers call them (src/recon_tmpl.c:990-1174, 2024-2049).
Value ranges follow tests/checkasm/mc.c:163-283, 482-771."""
import numpy as np
import pytest

import util
from dav1d_amd import api

from test_mc import _oracle_mc


def _need_ref():
    return util.default_oracle()


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_blend_matches_reference(ctx, bpc):
    """OBMC / inter-intra flow: mc PUT_TMP into the scratch arena, then blend / blend_h / blend_v onto dst."""
    oracle = util.default_oracle()
    rng = np.random.default_rng(900 + bpc)
    pd = util.pix_dtype(bpc)
    vis_w, vis_h = 200, 150
    ref = ctx.picture(vis_w, vis_h, api.LAYOUT_I400, bpc)
    refplane = rng.integers(0, 1 << bpc, size=ref.padded_shape(0)).astype(pd)
    ref.upload(0, refplane)
    DW = DH = 256
    dst = ctx.picture(DW, DH, api.LAYOUT_I400, bpc)
    dplane = rng.integers(0, 1 << bpc, size=dst.padded_shape(0)).astype(pd)
    dst.upload(0, dplane)
    want = dplane.copy()
    sp = dst.stride_px(0)
    n = 48
    mct = np.zeros(n, api.MC_TASK)
    cmp_ = np.zeros(n, api.COMP_TASK)
    masks = []
    want_tmp = []
    x = y = row_h = 0
    tmp_off = mask_off = 0
    k = 0
    for i in range(n):
        kind = 4 + i % 3
        if kind == 4:
            w = int(rng.choice([4, 8, 16, 32])); h = int(rng.choice([v for v in [4, 8, 16, 32] if w // 2 <= v <= w * 2]))
        elif kind == 5:
            w = int(rng.choice([2, 4, 8, 16, 32])); h = int(rng.choice([2, 4, 8, 16, 32, 64]))
        else:
            w = int(rng.choice([2, 4, 8, 16, 32, 64, 128])); h = int(rng.choice([2, 4, 8, 16, 32]))
        if x + w > DW:
            x = 0; y += row_h; row_h = 0
        if y + h > DH:
            break
        mct[k]["src_x"] = int(rng.choice([-3, int(rng.integers(4, 150)), vis_w - 5]))
        mct[k]["src_y"] = int(rng.choice([-2, int(rng.integers(4, 100)), vis_h - 3]))
        mct[k]["w"], mct[k]["h"] = w, h
        mct[k]["mx"], mct[k]["my"] = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        mct[k]["filter_2d"], mct[k]["kind"], mct[k]["dst_off"] = int(rng.integers(0, 10)), 2, tmp_off
        tmp = np.zeros((h, w), pd)
        _oracle_mc(oracle, bpc, refplane, vis_w, vis_h, mct[k], dst_block=tmp)
        want_tmp.append(tmp.ravel())
        blk = want[y:, x:]
        m = rng.integers(0, 65, size=w * h).astype(np.uint8)
        if kind == 4:
            oracle.call(bpc, "blend", 0, 0, blk.ctypes.data, want.strides[0], tmp, w, h, m)
        elif kind == 5:
            oracle.call(bpc, "blend_v", 0, 0, blk.ctypes.data, want.strides[0], tmp, w, h)
        else:
            oracle.call(bpc, "blend_h", 0, 0, blk.ctypes.data, want.strides[0], tmp, w, h)
        cmp_[k] = (y * sp + x, tmp_off, 0, mask_off, w, h, kind, 0, 0, 0, 0)
        masks.append(m)
        tmp_off += (w * h + 3) & ~3
        mask_off += w * h
        x += w
        row_h = max(row_h, h)
        k += 1
    mct, cmp_ = mct[:k], cmp_[:k]
    arena = ctx.buffer(tmp_off * pd().itemsize + 64)
    arena.zero()
    dmask = ctx.buffer_from(np.concatenate(masks))
    ctx.mc_batch(dst, [ref], mct, arena)
    got_tmp = arena.download(pd, tmp_off)
    assert np.array_equal(dst.download(0), dplane), "PUT_TMP must not touch the dst picture"
    for i in range(k):
        o = int(mct[i]["dst_off"])
        assert np.array_equal(got_tmp[o:o + len(want_tmp[i])], want_tmp[i]), ("put_tmp", tuple(mct[i]))
    ctx.comp_batch(dst, cmp_, arena, dmask)
    got = dst.download(0)
    bad = np.argwhere(got != want)
    assert not len(bad), ("blend", bad[:3], [tuple(t) for t in cmp_[:3]])
    for o in (ref, dst, arena, dmask):
        o.free()


def test_blend_h_before_blend_v(ctx):
    """obmc() order on one block: top neighbours (blend_h) first, left ones (blend_v) second; the areas overlap."""
    oracle = util.default_oracle()
    bpc = 8
    rng = np.random.default_rng(3)
    dst = ctx.picture(64, 64, api.LAYOUT_I400, bpc)
    dplane = rng.integers(0, 256, size=dst.padded_shape(0)).astype(np.uint8)
    dst.upload(0, dplane)
    want = dplane.copy()
    sp = dst.stride_px(0)
    w = h = 16
    tmps = [rng.integers(0, 256, size=w * h).astype(np.uint8) for _ in range(2)]
    blk = want[8:, 8:]
    oracle.call(bpc, "blend_h", 0, 0, blk.ctypes.data, want.strides[0], tmps[0], w, h)
    oracle.call(bpc, "blend_v", 0, 0, blk.ctypes.data, want.strides[0], tmps[1], w, h)
    tasks = np.zeros(2, api.COMP_TASK)
    tasks[0] = (8 * sp + 8, w * h, 0, 0, w, h, 5, 0, 0, 0, 0)     # blend_v listed first on purpose
    tasks[1] = (8 * sp + 8, 0, 0, 0, w, h, 6, 0, 0, 0, 0)
    arena = ctx.buffer_from(np.concatenate(tmps))
    ctx.comp_batch(dst, tasks, arena, None)
    assert np.array_equal(dst.download(0), want)
    dst.free(); arena.free()


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_warp_matches_reference(ctx, bpc):
    oracle = _need_ref()
    rng = np.random.default_rng(40 + bpc)
    pd = util.pix_dtype(bpc)
    bps = pd().itemsize
    vis_w, vis_h = 120, 90
    ref = ctx.picture(vis_w, vis_h, api.LAYOUT_I400, bpc)
    refplane = rng.integers(0, 1 << bpc, size=ref.padded_shape(0)).astype(pd)
    ref.upload(0, refplane)
    dst = ctx.picture(128, 128, api.LAYOUT_I400, bpc)
    dplane = rng.integers(0, 1 << bpc, size=dst.padded_shape(0)).astype(pd)
    dst.upload(0, dplane)
    want = dplane.copy()
    sp = dst.stride_px(0)
    n = 128
    tasks = np.zeros(n, api.WARP_TASK)
    want_prep = np.zeros(64 * 64, np.int16)
    for i in range(n):
        kind = i & 1
        bx, by = (i >> 1) % 8, (i >> 1) // 8
        dx = int(rng.choice([-9, -2, 1, int(rng.integers(3, vis_w - 12)), vis_w - 11, vis_w - 4, vis_w + 6]))
        dy = int(rng.choice([-7, 0, 2, int(rng.integers(3, vis_h - 12)), vis_h - 10, vis_h + 3]))
        mx, my = (int(rng.integers(0, 0x2000)) - 0xa00 for _ in range(2))
        abcd = (rng.integers(0, 0x2000, size=4) - 0xa00).astype(np.int16)
        # the driver's fetch (src/recon_tmpl.c:1146-1163)
        if dx < 3 or dx + 8 + 4 > vis_w or dy < 3 or dy + 8 + 4 > vis_h:
            emu = np.zeros((16, 32), pd)
            oracle.call(bpc, "emu_edge", 0, 0, 15, 15, vis_w, vis_h, dx - 3, dy - 3, emu.ctypes.data, 32 * bps,
                        refplane.ctypes.data, refplane.strides[0])
            src_ptr, src_stride = emu.ctypes.data + (32 * 3 + 3) * bps, 32 * bps
        else:
            src_ptr, src_stride = refplane.ctypes.data + dy * refplane.strides[0] + dx * bps, refplane.strides[0]
        if kind == 0:
            off = by * 8 * sp + bx * 8
            blk = want[by * 8:, bx * 8:]
            oracle.call(bpc, "warp8x8", 0, 0, blk.ctypes.data, want.strides[0], src_ptr, src_stride, abcd, mx, my)
        else:
            off = by * 8 * 64 + bx * 8
            oracle.call(bpc, "warp8x8t", 0, 0, want_prep[off:].ctypes.data, 64, src_ptr, src_stride, abcd, mx, my)
        tasks[i] = (off, dx, dy, mx, my, abcd, 64, kind, 0, 0, (0, 0, 0))
    prep = ctx.buffer(64 * 64 * 2)
    prep.zero()
    ctx.warp_batch(dst, [ref], tasks, prep)
    assert np.array_equal(dst.download(0), want)
    assert np.array_equal(prep.download(np.int16, 64 * 64), want_prep)
    for o in (ref, dst, prep):
        o.free()


@pytest.mark.parametrize("bpc", [8, 10, 12])
@pytest.mark.parametrize("kind", [0, 1], ids=["put", "prep"])
def test_mc_scaled_matches_reference(ctx, bpc, kind):
    oracle = _need_ref()
    rng = np.random.default_rng(60 + bpc * 2 + kind)
    pd = util.pix_dtype(bpc)
    bps = pd().itemsize
    vis_w, vis_h = 300, 280
    ref = ctx.picture(vis_w, vis_h, api.LAYOUT_I400, bpc)
    refplane = rng.integers(0, 1 << bpc, size=ref.padded_shape(0)).astype(pd)
    ref.upload(0, refplane)
    DW = DH = 512
    dst = ctx.picture(DW, DH, api.LAYOUT_I400, bpc)
    dplane = rng.integers(0, 1 << bpc, size=dst.padded_shape(0)).astype(pd)
    dst.upload(0, dplane)
    want = dplane.copy()
    sp = dst.stride_px(0)
    n = 40 if ctx.backend == "emu" else 160
    tasks = np.zeros(n, api.MC_SCALED_TASK)
    want_prep = []
    x = y = row_h = 0
    prep_off = 0
    k = 0
    for i in range(n):
        w = int(rng.choice([2, 4, 8, 16, 32, 64, 128], p=[.1, .2, .25, .2, .12, .08, .05]))
        h = int(rng.choice([v for v in [2, 4, 8, 16, 32, 64, 128] if max(w // 4, 2) <= v <= min(w * 4, 128)]))
        if kind == 1:
            w = max(w, 4)
        if x + w > DW:
            x = 0; y += row_h; row_h = 0
        if y + h > DH:
            break
        mx, my = int(rng.integers(0, 1024)), int(rng.integers(0, 1024))
        step_x = int(rng.choice([512, 1024, 2048, int(rng.integers(1, 2049))]))
        step_y = int(rng.choice([512, 1024, 2048, int(rng.integers(1, 2049))]))
        left = int(rng.choice([-20, -2, 1, int(rng.integers(3, vis_w)), vis_w - 5, vis_w + 3]))
        top = int(rng.choice([-11, 0, 2, int(rng.integers(3, vis_h)), vis_h - 3, vis_h + 8]))
        right = left + ((mx + (w - 1) * step_x) >> 10) + 1
        bottom = top + ((my + (h - 1) * step_y) >> 10) + 1
        if left < 3 or top < 3 or right + 4 > vis_w or bottom + 4 > vis_h:
            emu = np.zeros((bottom - top + 8, 320), pd)
            oracle.call(bpc, "emu_edge", 0, 0, right - left + 7, bottom - top + 7, vis_w, vis_h, left - 3, top - 3,
                        emu.ctypes.data, 320 * bps, refplane.ctypes.data, refplane.strides[0])
            src_ptr, src_stride = emu.ctypes.data + (320 * 3 + 3) * bps, 320 * bps
        else:
            src_ptr, src_stride = refplane.ctypes.data + top * refplane.strides[0] + left * bps, refplane.strides[0]
        f = int(rng.integers(0, 10))
        if kind == 0:
            off = y * sp + x
            blk = want[y:, x:]
            oracle.call(bpc, "mc_scaled", f, 0, blk.ctypes.data, want.strides[0], src_ptr, src_stride, w, h, mx, my, step_x, step_y)
        else:
            off = prep_off
            tmp = np.zeros(w * h, np.int16)
            oracle.call(bpc, "mct_scaled", f, 0, tmp.ctypes.data, src_ptr, src_stride, w, h, mx, my, step_x, step_y)
            want_prep.append(tmp)
            prep_off += w * h
        tasks[k] = (off, left, top, mx, my, step_x, step_y, w, h, f, kind, 0, 0, (0, 0))
        x += w
        row_h = max(row_h, h)
        k += 1
    tasks = tasks[:k]
    assert k > 20
    prep = ctx.buffer(max(prep_off, 8) * 2)
    prep.zero()
    ctx.mc_scaled_batch(dst, [ref], tasks, prep)
    got = dst.download(0)
    bad = np.argwhere(got != want)
    assert not len(bad), ("mc_scaled", bad[:3], got[tuple(bad[0])], want[tuple(bad[0])])
    if kind == 1:
        got_prep = prep.download(np.int16, prep_off)
        wp = np.concatenate(want_prep)
        b = np.flatnonzero(got_prep != wp)
        if len(b):
            hit = [tuple(t) for t in tasks if t["dst_off"] <= b[0] < t["dst_off"] + int(t["w"]) * int(t["h"])]
            raise AssertionError("mct_scaled mismatch at %d: got %d want %d task %s" % (b[0], got_prep[b[0]], wp[b[0]], hit[:1]))
    for o in (ref, dst, prep):
        o.free()


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_resize_matches_reference(ctx, bpc):
    oracle = _need_ref()
    rng = np.random.default_rng(11 + bpc)
    pd = util.pix_dtype(bpc)
    for it in range(4):
        w_den = 9 + int(rng.integers(0, 8))
        src_w = 16 + int(rng.integers(0, 512 - 16 + 1))
        dst_w = w_den * src_w >> 3
        H = 40
        dx = ((src_w << 14) + (dst_w >> 1)) // dst_w
        err = dst_w * dx - (src_w << 14)
        # get_upscale_x0, reference src/decode.c:3321-3325 (C division truncates toward zero)
        num = -((dst_w - src_w) << 13) + (dst_w >> 1)
        q = abs(num) // dst_w * (1 if num >= 0 else -1)
        e2 = abs(err) // 2 * (1 if err >= 0 else -1)
        mx0 = (q + 128 - e2) & 0x3fff
        src = ctx.picture(src_w, H, api.LAYOUT_I400, bpc)
        splane = rng.integers(0, 1 << bpc, size=src.padded_shape(0)).astype(pd)
        src.upload(0, splane)
        dst = ctx.picture(dst_w, H, api.LAYOUT_I400, bpc)
        dplane = rng.integers(0, 1 << bpc, size=dst.padded_shape(0)).astype(pd)
        dst.upload(0, dplane)
        want = dplane.copy()
        y0, h = (0, H) if it & 1 else (8, 24)
        oracle.call(bpc, "resize", 0, 0, want[y0:].ctypes.data, want.strides[0], splane[y0:].ctypes.data, splane.strides[0],
                    dst_w, h, src_w, dx, mx0)
        ctx.resize(dst, src, 0, dst_w, y0, h, src_w, dx, mx0)
        assert np.array_equal(dst.download(0), want), (src_w, dst_w, dx, mx0)
        src.free(); dst.free()


@pytest.mark.parametrize("bpc", [8, 10])
def test_emu_edge_matches_reference(ctx, bpc):
    """tests/checkasm/mc.c:663-715: every block size against every side of the plane."""
    oracle = util.default_oracle()
    rng = np.random.default_rng(21 + bpc)
    pd = util.pix_dtype(bpc)
    bps = pd().itemsize
    iw, ih = 70, 50
    plane = rng.integers(0, 1 << bpc, size=(ih, 80)).astype(pd)
    dref = ctx.buffer_from(plane)
    for bw, bh in [(4, 4), (15, 15), (71, 9), (135, 135), (23, 135)]:
        for x, y in [(-bw - 3, 5), (-3, -2), (iw - 5, ih - 2), (iw + 4, 10), (20, -bh - 1), (10, ih + 7), (12, 9), (-40, -60)]:
            want = np.zeros((bh, 192), pd)
            oracle.call(bpc, "emu_edge", 0, 0, bw, bh, iw, ih, x, y, want.ctypes.data, 192 * bps, plane.ctypes.data, plane.strides[0])
            ddst = ctx.buffer(bh * 192 * bps)
            ddst.zero()
            ctx.emu_edge(bpc, bw, bh, iw, ih, x, y, ddst, 192 * bps, dref, plane.strides[0])
            got = ddst.download(pd, bh * 192).reshape(bh, 192)
            assert np.array_equal(got[:, :bw], want[:, :bw]), (bw, bh, x, y)
            ddst.free()
    dref.free()
