"""itxfm_add parity: HIP kernels (through the C ABI) vs the reference C functions.

Input generation follows tests/checkasm/itx.c:185-305 (forward float transform of a random
residual, random eob per sub-size class so dc-only / partial paths are hit); both the pixels
and the zeroed coefficient slabs are compared byte for byte."""
import numpy as np
import pytest

import util
from dav1d_amd import api


def _build_case(rng, bpc, txs, per_type, layout=api.LAYOUT_I400, W=256, H=256):
    """Tile a WxH plane with blocks of the requested tx sizes; returns tasks, coef arena, block list."""
    tasks, coefs, blocks = [], [], []
    cf_off = 0
    x = y = 0
    row_h = 0
    for tx in txs:
        w, h = util.TX_W[tx], util.TX_H[tx]
        for txtp in util.legal_txtps(tx):
            for rep in range(per_type):
                for subsh in range(1 if txtp else 0, util.subsh_max(tx)):
                    if txtp == util.WHT_WHT and subsh > 1:
                        continue
                    if x + w > W:
                        x = 0; y += row_h; row_h = 0
                    if y + h > H:
                        return tasks, coefs, blocks, True
                    cf, eob = util.gen_itx_coefs(rng, tx, txtp, bpc, subsh)
                    tasks.append((x, y, cf_off, eob, tx, txtp))
                    coefs.append(cf)
                    blocks.append((x, y, w, h))
                    cf_off += len(cf)
                    x += w
                    row_h = max(row_h, h)
    return tasks, coefs, blocks, False


def _run_case(ctx, oracle, bpc, txs, per_type, seed, packed=False):
    rng = np.random.default_rng(seed)
    W = H = 256
    tasks, coefs, blocks, trunc = _build_case(rng, bpc, txs, per_type, W=W, H=H)
    assert tasks and not trunc
    pd = util.pix_dtype(bpc)
    plane = rng.integers(0, 1 << bpc, size=(H, W)).astype(pd)
    arena = np.concatenate(coefs)
    # ---- oracle, block by block, on host copies
    ref_plane = plane.copy()
    ref_arena = arena.copy()
    for (x, y, cf_off, eob, tx, txtp) in tasks:
        n = len(coefs[0]) * 0 + min(util.TX_W[tx], 32) * min(util.TX_H[tx], 32)
        dst = ref_plane[y:, x:]
        oracle.call(bpc, "itxfm_add", tx, txtp, dst.ctypes.data, ref_plane.strides[0],
                    ref_arena[cf_off:cf_off + n].ctypes.data, eob)
    # ---- backend through the C ABI
    pic = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    pic.upload(0, plane)
    t = np.zeros(len(tasks), api.ITX_TASK)
    sp = pic.stride_px(0)
    if packed:
        # sparse wire format: only the eob + 1 coefficients of every block, in decode order, back to back
        parts, off = [], 0
        for i, (x, y, cf_off, eob, tx, txtp) in enumerate(tasks):
            n = min(util.TX_W[tx], 32) * min(util.TX_H[tx], 32)
            pk = util.pack_coefs(tx, txtp, arena[cf_off:cf_off + n], eob)
            t[i] = (y * sp + x, off, eob, tx, txtp, 0, 1, (0, 0))
            parts.append(pk)
            off += len(pk)
        arena = np.concatenate(parts)
        ref_arena = arena.copy()              # read-only for the backend
    else:
        for i, (x, y, cf_off, eob, tx, txtp) in enumerate(tasks):
            t[i] = (y * sp + x, cf_off, eob, tx, txtp, 0, 0, (0, 0))
    dcoef = ctx.buffer_from(arena)
    perm = rng.permutation(len(t))            # any order must give the same result
    ctx.itx_add_batch(pic, t[perm], dcoef)
    out = pic.download(0)
    out_arena = dcoef.download(arena.dtype, len(arena))
    pic.free(); dcoef.free()
    bad = np.argwhere(out != ref_plane)
    if len(bad):
        yy, xx = bad[0]
        blk = [b for b in zip(tasks, blocks) if b[1][0] <= xx < b[1][0] + b[1][2] and b[1][1] <= yy < b[1][1] + b[1][3]]
        raise AssertionError("pixel mismatch at (%d,%d): got %d want %d; block %s" %
                             (xx, yy, out[yy, xx], ref_plane[yy, xx], blk[:1]))
    assert np.array_equal(out_arena, ref_arena), "coefficient slabs not zeroed like the reference" if not packed else "packed arena modified"


@pytest.mark.parametrize("bpc", [8, 10, 12])
@pytest.mark.parametrize("tx", list(range(19)), ids=util.TX_NAMES)
def test_itxfm_add_matches_reference(ctx, bpc, tx):
    per_type = 1 if ctx.backend == "emu" else 2
    _run_case(ctx, util.default_oracle(), bpc, [tx], per_type, seed=1000 + tx * 3 + bpc)


@pytest.mark.parametrize("bpc", [8, 10, 12])
@pytest.mark.parametrize("tx", list(range(19)), ids=util.TX_NAMES)
def test_itxfm_add_from_packed_coefficients(ctx, bpc, tx):
    """DAV1D_HIP_ITX_PACKED: the same blocks fed as eob + 1 scan-order values each give the same pixels."""
    _run_case(ctx, util.default_oracle(), bpc, [tx], 1, seed=5000 + tx * 3 + bpc, packed=True)


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_itxfm_add_with_coefficients_at_the_limits(ctx, bpc):
    """Every transform size and type fed with coefficients at and around the largest magnitudes the entropy decoder can hand over
    (cf_max, src/recon_tmpl.c:decode_coefs: 0x7fff / 0x1ffff / 0x7ffff) in random sign patterns, dense and sparse.  The reference
    clamps on the way in and after every add / sub stage; the kernels multiply on the 24-bit unit, which is exact only inside that
    range (the emulated build traps on an operand outside it), so this is the case that would tell the two apart."""
    oracle = util.default_oracle()
    rng = np.random.default_rng(7700 + bpc)
    cf_max = (1 << (15 if bpc == 8 else bpc + 7)) - 1
    cdt = np.int16 if bpc == 8 else np.int32
    W = H = 512
    pd = util.pix_dtype(bpc)
    plane = rng.integers(0, 1 << bpc, size=(H, W)).astype(pd)
    tasks, coefs = [], []
    x = y = row_h = cf_off = 0
    for tx in range(19):
        w, h = util.TX_W[tx], util.TX_H[tx]
        n = min(w, 32) * min(h, 32)
        for txtp in util.legal_txtps(tx):
            for kind in range(3):
                if x + w > W:
                    x = 0; y += row_h; row_h = 0
                assert y + h <= H
                mag = np.array([cf_max, cf_max, cf_max - 1, cf_max // 2 + 1, 1])[rng.integers(0, 5, n)]
                cf = (mag * rng.choice([-1, 1], n)).astype(cdt)
                if kind == 1:
                    cf[rng.random(n) < 0.7] = 0
                elif kind == 2:
                    cf[1:] = 0
                if txtp == util.WHT_WHT:
                    cf = (cf >> (7 if bpc == 8 else bpc - 1)).astype(cdt)      # lossless blocks carry residuals, not scaled coefficients
                nz = np.flatnonzero(cf)
                eob = int(nz[-1]) if len(nz) else 0
                if kind != 2:
                    eob = n - 1           # dense: no shortcut by eob class
                tasks.append((x, y, cf_off, eob, tx, txtp))
                coefs.append(cf)
                cf_off += n
                x += w
                row_h = max(row_h, h)
    arena = np.concatenate(coefs)
    ref_plane, ref_arena = plane.copy(), arena.copy()
    for (bx, by, off, eob, tx, txtp) in tasks:
        n = min(util.TX_W[tx], 32) * min(util.TX_H[tx], 32)
        oracle.call(bpc, "itxfm_add", tx, txtp, ref_plane[by:, bx:].ctypes.data, ref_plane.strides[0], ref_arena[off:off + n].ctypes.data, eob)
    pic = ctx.picture(W, H, api.LAYOUT_I400, bpc)
    pic.upload(0, plane)
    sp = pic.stride_px(0)
    t = np.zeros(len(tasks), api.ITX_TASK)
    for i, (bx, by, off, eob, tx, txtp) in enumerate(tasks):
        t[i] = (by * sp + bx, off, eob, tx, txtp, 0, 0, (0, 0))
    dcoef = ctx.buffer_from(arena)
    ctx.itx_add_batch(pic, t, dcoef)
    out = pic.download(0)
    pic.free(); dcoef.free()
    bad = np.argwhere(out != ref_plane)
    if len(bad):
        yy, xx = bad[0]
        blk = [b for b in tasks if b[0] <= xx < b[0] + util.TX_W[b[4]] and b[1] <= yy < b[1] + util.TX_H[b[4]]]
        raise AssertionError("pixel mismatch at (%d,%d): got %d want %d; block %s (%d px differ)" % (xx, yy, out[yy, xx], ref_plane[yy, xx], blk[:1], len(bad)))
