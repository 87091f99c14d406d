"""Deterministic oracle-only cases whose outputs are pinned as SHA-256 digests in
tests/golden/dsp_golden.json.  The digests were produced by tests/golden/make_golden.py from the
reference's own C functions (oracle/_ref); test_oracle.py replays the same cases through
oracle/port (and through oracle/_ref again when it is present) and compares digests, so the C
restatement stays pinned even where /root/reference and oracle/_ref are absent."""
import hashlib

import numpy as np

import util
import test_mc


def _digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def itx_case(oracle, bpc, tx):
    rng = np.random.default_rng(4242 + 31 * tx + bpc)
    w, h = util.TX_W[tx], util.TX_H[tx]
    n = min(w, 32) * min(h, 32)
    pd = util.pix_dtype(bpc)
    outs = []
    for txtp in util.legal_txtps(tx):
        for subsh in range(1 if txtp else 0, util.subsh_max(tx)):
            if txtp == util.WHT_WHT and subsh > 1:
                continue
            cf, eob = util.gen_itx_coefs(rng, tx, txtp, bpc, subsh)
            dst = rng.integers(0, 1 << bpc, size=(h, w + 3)).astype(pd)
            oracle.call(bpc, "itxfm_add", tx, txtp, dst.ctypes.data, dst.strides[0], cf, eob)
            outs += [dst, cf[:n]]
    return _digest(*outs)


def mc_case(oracle, bpc, kind):
    rng = np.random.default_rng(900 + bpc * 2 + kind)
    vis_w, vis_h = 200, 150
    pd = util.pix_dtype(bpc)
    refplane = rng.integers(0, 1 << bpc, size=(256, 256)).astype(pd)
    tasks, pos, prep_sz = test_mc._gen_tasks(rng, 120, vis_w, vis_h, 512, 512, kind, test_mc.SIZES_PUT)
    plane = rng.integers(0, 1 << bpc, size=(512, 512)).astype(pd)
    prep = np.zeros(max(prep_sz, 1), np.int16)
    for i, t in enumerate(tasks):
        if kind == 0:
            x, y = pos[i]
            test_mc._oracle_mc(oracle, bpc, refplane, vis_w, vis_h, t, dst_block=plane[y:, x:])
        else:
            test_mc._oracle_mc(oracle, bpc, refplane, vis_w, vis_h, t, tmp=prep[pos[i]:])
    return _digest(plane, prep)


def comp_case(oracle, bpc):
    rng = np.random.default_rng(77 + bpc)
    pd = util.pix_dtype(bpc)
    bias = 8192 if bpc > 8 else 0
    ib = 4 if bpc == 8 else 14 - bpc
    outs = []
    for kind, ss in [(0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 2)] * 4:
        w = int(rng.choice([4, 8, 16, 32, 64, 128]))
        h = int(rng.choice([v for v in [4, 8, 16, 32, 64, 128] if w // 4 <= v <= w * 4]))
        t1 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
        t2 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
        dst = rng.integers(0, 1 << bpc, size=(h, w)).astype(pd)
        m = rng.integers(0, 65, size=w * h).astype(np.uint8)
        mo = np.zeros(w * h, np.uint8)
        if kind == 0:
            oracle.call(bpc, "avg", 0, 0, dst, dst.strides[0], t1, t2, w, h)
        elif kind == 1:
            oracle.call(bpc, "w_avg", 0, 0, dst, dst.strides[0], t1, t2, w, h, int(rng.integers(1, 16)))
        elif kind == 2:
            oracle.call(bpc, "mask", 0, 0, dst, dst.strides[0], t1, t2, w, h, m)
        else:
            oracle.call(bpc, "w_mask", ss, 0, dst, dst.strides[0], t1, t2, w, h, mo, int(rng.integers(0, 2)))
            n = (w >> (1 if ss else 0)) * (h >> (1 if ss == 2 else 0))
            outs.append(mo[:n])
        outs.append(dst)
    return _digest(*outs)


def blend_case(oracle, bpc):
    rng = np.random.default_rng(1234 + bpc)
    pd = util.pix_dtype(bpc)
    outs = []
    for fam in ("blend", "blend_v", "blend_h"):
        for w in (4, 8, 16, 32):       # blend_v: w up to 32, blend_h: h up to 32 (obmc mask table)
            for h in (4, 8, 16, 32):
                dst = rng.integers(0, 1 << bpc, size=(h, w + 5)).astype(pd)
                tmp = rng.integers(0, 1 << bpc, size=(h, w)).astype(pd)
                if fam == "blend":
                    m = rng.integers(0, 65, size=w * h).astype(np.uint8)
                    oracle.call(bpc, fam, 0, 0, dst, dst.strides[0], tmp, w, h, m)
                else:
                    oracle.call(bpc, fam, 0, 0, dst, dst.strides[0], tmp, w, h)
                outs.append(dst)
    return _digest(*outs)


# ---------------------------------------------------------------- post filters, intra prediction, film grain

def _frame(rng, bpc, h, w):
    return rng.integers(0, 1 << bpc, size=(h, w)).astype(util.pix_dtype(bpc))


def _ptr(arr, *idx):
    return arr.ctypes.data + sum(int(i) * int(s) for i, s in zip(idx, arr.strides))


def _lf_lut(sharp):
    """Av1FilterLUT as bytes: e[64], i[64], sharp[2] (dav1d_calc_eih, reference src/lf_mask.c:385-410)."""
    buf = np.zeros(144, np.uint8)
    for level in range(64):
        limit = level
        if sharp > 0:
            limit >>= (sharp + 3) >> 2
            limit = min(limit, 9 - sharp)
        limit = max(limit, 1)
        buf[64 + level] = limit
        buf[level] = 2 * (level + 2) + limit
    return buf


def lf_case(oracle, bpc):
    rng = np.random.default_rng(40 + bpc)
    pd = util.pix_dtype(bpc)
    bd8 = bpc - 8
    outs = []
    for rep in range(3):
        lut = _lf_lut(int(rng.integers(0, 8)))
        for chroma in (0, 1):
            for d in (0, 1):
                base = np.kron(rng.integers(-12, 13, size=(10, 10)) << bd8, np.ones((16, 16), np.int64))
                amp = np.kron(rng.choice([0, 1, 2, 6], size=(10, 10)) << bd8, np.ones((16, 16), np.int64))
                plane = np.clip((1 << (bpc - 1)) + base + rng.integers(-1, 2, size=(160, 160)) * amp, 0, (1 << bpc) - 1).astype(pd)
                lvl = rng.integers(0, 64, size=(40, 40, 4)).astype(np.uint8)
                lvl[rng.random((40, 40)) < 0.2] = 0
                m = rng.integers(0, 1 << 32, size=3, dtype=np.uint64).astype(np.uint32)
                m[0] &= np.uint32(0x00ffffff)
                m[1] &= m[0]
                m[2] &= m[1]
                if chroma:
                    m[2] = 0
                oracle.call(bpc, "loop_filter_sb", chroma, d, _ptr(plane, 16, 16), plane.strides[0], np.ascontiguousarray(m),
                            _ptr(lvl, 4, 4), 40, lut, 24)
                outs.append(plane)
    return _digest(*outs)


def cdef_case(oracle, bpc):
    import ctypes as C
    rng = np.random.default_rng(50 + bpc)
    pd = util.pix_dtype(bpc)
    bd8 = bpc - 8
    outs = []
    for it in range(12):
        if it & 1:
            ramp = np.arange(8)[None, :] * int(rng.integers(-6, 7)) + np.arange(8)[:, None] * int(rng.integers(-6, 7))
            blk = np.clip(int(rng.integers(0, 1 << bpc)) + (rng.integers(-8, 9, size=(8, 8)) + ramp) * (1 << bd8), 0, (1 << bpc) - 1).astype(pd)
        else:
            blk = _frame(rng, bpc, 8, 8)
        var = C.c_uint(0)
        d = oracle.call(bpc, "cdef_dir", 0, 0, blk, blk.strides[0], C.addressof(var))
        outs.append(np.array([d, var.value], np.int64))
    for fb, (w, h) in enumerate([(8, 8), (4, 8), (4, 4)]):
        for edges in range(16):
            for rep in range(2):
                plane = np.clip((1 << (bpc - 1)) + rng.integers(-(20 << bd8), (20 << bd8) + 1, size=(16, 16)), 0, (1 << bpc) - 1).astype(pd)
                left, top, bot = _frame(rng, bpc, 8, 2), _frame(rng, bpc, 2, 16), _frame(rng, bpc, 2, 16)
                pri = int(rng.integers(0, 16)) << bd8
                sec = (1 << int(rng.integers(0, 3))) << bd8 if rng.integers(0, 4) else 0
                if not pri and not sec:
                    pri = 3 << bd8
                oracle.call(bpc, "cdef_fb", fb, 0, _ptr(plane, 4, 4), plane.strides[0], left, _ptr(top, 0, 4), _ptr(bot, 0, 4), pri, sec,
                            int(rng.integers(0, 8)), int(rng.integers(3, 7)) + bd8, edges)
                outs.append(plane)
    return _digest(*outs)


SGR_PARAMS = [(140, 3236), (112, 2158), (93, 1618), (80, 1438), (70, 1295), (58, 1177), (47, 1079), (37, 996), (30, 925), (25, 863),
              (0, 2589), (0, 1618), (0, 1177), (0, 925), (56, 0), (22, 0)]        # dav1d_sgr_params, reference src/tables.c:415-420


def lr_case(oracle, bpc):
    import ctypes as C
    from test_lr import wiener_params
    rng = np.random.default_rng(60 + bpc)
    outs = []
    prm = (C.c_int16 * 16)()
    for fam, idx in [("wiener", 0), ("wiener", 1), ("sgr", 0), ("sgr", 1), ("sgr", 2)]:
        for edges in list(range(16)) + [15, 15]:
            w, h = int(rng.integers(1, 100)), int(rng.integers(1, 65))
            plane, left, lpf = _frame(rng, bpc, 80, 128), _frame(rng, bpc, 64, 4), _frame(rng, bpc, 10, 128)
            if fam == "wiener":
                f = wiener_params(rng, bpc, idx == 1)
                for k in range(16):
                    prm[k] = int(f[k // 8][k % 8])
            else:
                sets = [s for s in range(16) if (bool(SGR_PARAMS[s][0]) + 2 * bool(SGR_PARAMS[s][1]) - 1) == idx]
                s0, s1 = SGR_PARAMS[int(rng.choice(sets))]
                w0 = int(rng.integers(-96, 32))
                w1 = 160 - int(rng.integers(0, 128)) - w0
                if idx == 0:
                    w1 = 0
                if idx == 1:
                    w0, w1 = 0, int(rng.integers(-32, 96))
                u = np.array([s0, s1], np.uint32).view(np.int16)
                for k in range(4):
                    prm[k] = int(u[k])
                prm[4], prm[5] = w0, w1
            oracle.call(bpc, fam, idx, 0, _ptr(plane, 8, 8), plane.strides[0], left, _ptr(lpf, 0, 8), w, h, C.addressof(prm), edges)
            outs.append(plane[8:8 + h, 8:8 + w])
    return _digest(*outs)


def ipred_case(oracle, bpc):
    rng = np.random.default_rng(30 + bpc)
    pd = util.pix_dtype(bpc)
    bps = pd().itemsize
    outs = []
    sizes = [(w, h) for w in (4, 8, 16, 32, 64) for h in (4, 8, 16, 32, 64) if max(w, h) <= 4 * min(w, h)]
    for mode in range(14):
        for w, h in sizes:
            if mode == 13 and (w > 32 or h > 32):
                continue
            for rep in range(2):
                edge = _frame(rng, bpc, 1, 257).ravel()
                if rep:         # smooth edges keep the directional interpolation away from saturation
                    edge = np.clip((1 << (bpc - 1)) + np.cumsum(rng.integers(-3, 4, size=257)) * (1 << (bpc - 8)), 0, (1 << bpc) - 1).astype(pd)
                a = 0
                if 6 <= mode <= 8:
                    a = (90 * (mode - 6) + int(rng.choice(np.arange(3, 90, 3)))) | (int(rng.integers(0, 4)) << 9)
                elif mode == 13:
                    a = int(rng.integers(0, 5))
                dst = np.zeros((h, w + 1), pd)
                oracle.call(bpc, "intra_pred", mode, 0, dst, dst.strides[0], edge.ctypes.data + 128 * bps, w, h, a,
                            int(rng.integers(1, w + 1)), int(rng.integers(1, h + 1)))
                outs.append(dst)
    for layout in range(3):
        ss_hor, ss_ver = layout < 2, layout == 0
        for cw, ch in [(4, 4), (4, 8), (8, 16), (16, 8), (32, 32), (16, 32)]:
            for w_pad, h_pad in [(0, 0), (cw // 4 - 1, 0), (0, ch // 4 - 1), ((cw // 4) // 2, (ch // 4) // 2)]:
                luma = _frame(rng, bpc, ch << ss_ver, (cw << ss_hor) + 2)
                ac = np.zeros(cw * ch, np.int16)
                oracle.call(bpc, "cfl_ac", layout, 0, ac, luma, luma.strides[0], w_pad, h_pad, cw, ch)
                outs.append(ac)
    for mode in (0, 3, 4, 5):
        for w, h in [(4, 4), (8, 16), (32, 8), (16, 16), (4, 16)]:
            edge = _frame(rng, bpc, 1, 257).ravel()
            ac = rng.integers(-(1 << (bpc + 2)), 1 << (bpc + 2), size=w * h).astype(np.int16)
            dst = np.zeros((h, w), pd)
            oracle.call(bpc, "cfl_pred", mode, 0, dst, dst.strides[0], edge.ctypes.data + 128 * bps, w, h, ac, int(rng.integers(-16, 17)))
            outs.append(dst)
    for w, h in [(4, 4), (8, 8), (64, 16), (16, 64)]:
        pal = _frame(rng, bpc, 1, 8).ravel()
        idx = (rng.integers(0, 8, size=w * h // 2) | (rng.integers(0, 8, size=w * h // 2) << 4)).astype(np.uint8)
        dst = np.zeros((h, w), pd)
        oracle.call(bpc, "pal_pred", 0, 0, dst, dst.strides[0], pal, idx, w, h)
        outs.append(dst)
    return _digest(*outs)


def fg_case(oracle, bpc):
    import ctypes as C
    from test_filmgrain import random_fg
    rng = np.random.default_rng(70 + bpc)
    pd = util.pix_dtype(bpc)
    ent = np.int8 if bpc == 8 else np.int16
    outs = []
    for variant in range(4):
        d = random_fg(rng, bpc, variant)
        gy = np.zeros((74, 82), ent)
        oracle.call(bpc, "generate_grain_y", 0, 0, gy, C.addressof(d))
        outs.append(gy)
        scaling = rng.integers(0, 256, size=1 << bpc).astype(np.uint8)
        pw = 100
        for row_num, bh in [(0, 32), (3, 32), (5, 17)]:
            src, dst = _frame(rng, bpc, 32, 128), np.zeros((32, 128), pd)
            oracle.call(bpc, "fgy_32x32xn", 0, 0, dst, src, src.strides[0], C.addressof(d), pw, scaling, gy, bh, row_num)
            outs.append(dst)
        for layout in range(3):
            sx, sy = layout < 2, layout == 0
            for uv in range(2):
                g = np.zeros((74, 82), ent)
                oracle.call(bpc, "generate_grain_uv", layout, 0, g, gy, C.addressof(d), uv)
                outs.append(g)
                for row_num, lbh in [(0, 32), (2, 32), (4, 18)]:
                    luma, csrc, dst = _frame(rng, bpc, 32, 128), _frame(rng, bpc, 32, 128), np.zeros((32, 128), pd)
                    oracle.call(bpc, "fguv_32x32xn", layout, 0, dst, csrc, csrc.strides[0], C.addressof(d), pw >> sx, scaling, g,
                                (lbh + sy) >> sy, row_num, luma, luma.strides[0], uv, int(rng.integers(0, 2)))
                    outs.append(dst)
    return _digest(*outs)


def mcx_case(oracle, bpc):
    """warp8x8 / warp8x8t, scaled put / prep, resize (value ranges of tests/checkasm/mc.c:163-283, 565-771)."""
    rng = np.random.default_rng(4100 + bpc)
    pd = util.pix_dtype(bpc)
    bps = pd().itemsize
    src = _frame(rng, bpc, 300, 320)
    outs = []
    for it in range(24):
        abcd = (rng.integers(0, 0x2000, size=4) - 0xa00).astype(np.int16)
        mx, my = int(rng.integers(0, 0x2000)) - 0xa00, int(rng.integers(0, 0x2000)) - 0xa00
        sp = _ptr(src, 20 + it, 24 + 3 * it)
        d = np.zeros((8, 8), pd)
        oracle.call(bpc, "warp8x8", 0, 0, d, 8 * bps, sp, src.strides[0], abcd, mx, my)
        t = np.zeros((8, 8), np.int16)
        oracle.call(bpc, "warp8x8t", 0, 0, t, 8, sp, src.strides[0], abcd, mx, my)
        outs += [d, t]
    for f in range(10):
        for w, h in [(2, 2), (4, 8), (8, 4), (16, 32), (64, 16), (128, 8)]:
            mx, my = int(rng.integers(0, 1024)), int(rng.integers(0, 1024))
            dx = int(rng.choice([512, 1024, 2048, int(rng.integers(1, 2049))]))
            dy = int(rng.choice([512, 1024, 2048, int(rng.integers(1, 2049))]))
            sp = _ptr(src, 8, 8)
            d = np.zeros((h, w + 1), pd)
            oracle.call(bpc, "mc_scaled", f, 0, d, d.strides[0], sp, src.strides[0], w, h, mx, my, dx, dy)
            outs.append(d)
            if w >= 4:
                t = np.zeros(w * h, np.int16)
                oracle.call(bpc, "mct_scaled", f, 0, t, sp, src.strides[0], w, h, mx, my, dx, dy)
                outs.append(t)
    for it in range(6):
        w_den = 9 + int(rng.integers(0, 8))
        src_w = 16 + int(rng.integers(0, 300 - 16 + 1))
        dst_w = w_den * src_w >> 3
        dx = ((src_w << 14) + (dst_w >> 1)) // dst_w
        d = np.zeros((20, dst_w + 2), pd)
        oracle.call(bpc, "resize", 0, 0, d, d.strides[0], src, src.strides[0], dst_w, 20, src_w, dx, int(rng.integers(0, 0x4000)))
        outs.append(d)
    return _digest(*outs)


def all_cases():
    """name -> callable(oracle)"""
    cases = {}
    for bpc in (8, 10, 12):
        for tx in range(19):
            cases["itx/%dbpc/%s" % (bpc, util.TX_NAMES[tx])] = (lambda o, b=bpc, t=tx: itx_case(o, b, t))
        for kind, nm in ((0, "put"), (1, "prep")):
            cases["mc/%dbpc/%s" % (bpc, nm)] = (lambda o, b=bpc, k=kind: mc_case(o, b, k))
        cases["comp/%dbpc" % bpc] = (lambda o, b=bpc: comp_case(o, b))
        cases["blend/%dbpc" % bpc] = (lambda o, b=bpc: blend_case(o, b))
        cases["mcx/%dbpc" % bpc] = (lambda o, b=bpc: mcx_case(o, b))
        cases["lf/%dbpc" % bpc] = (lambda o, b=bpc: lf_case(o, b))
        cases["cdef/%dbpc" % bpc] = (lambda o, b=bpc: cdef_case(o, b))
        cases["lr/%dbpc" % bpc] = (lambda o, b=bpc: lr_case(o, b))
        cases["ipred/%dbpc" % bpc] = (lambda o, b=bpc: ipred_case(o, b))
        cases["fg/%dbpc" % bpc] = (lambda o, b=bpc: fg_case(o, b))
    return cases
