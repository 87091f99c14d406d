"""The kernel-level drop-in: dav1d_hip_dsp_init_{8,16}bpc fills a table laid out like the reference's Dav1dDSPContext
(src/internal.h:62-70) with reference-signature functions.  These tests (1) pin that layout against the table of the
reference build itself, slot by slot, and (2) call every family through both tables with identical host arguments,
the way tests/checkasm drives a DSP implementation against the C one."""
import ctypes as C

import numpy as np
import pytest

import util
from test_filmgrain import random_fg
from test_loopfilter import LutStruct, make_lut, structured_plane
from test_lr import wiener_params

N_SLOTS = 421
# first slot of every member, in the reference's declaration order (fg, ipred, mc, itx, lf, cdef, lr)
BASE = {
    "generate_grain_y": 0, "generate_grain_uv": 1, "fgy_32x32xn": 4, "fguv_32x32xn": 5,
    "intra_pred": 8, "cfl_ac": 22, "cfl_pred": 25, "pal_pred": 31,
    "mc": 32, "mc_scaled": 42, "mct": 52, "mct_scaled": 62, "avg": 72, "w_avg": 73, "mask": 74, "w_mask": 75,
    "blend": 78, "blend_v": 79, "blend_h": 80, "warp8x8": 81, "warp8x8t": 82, "emu_edge": 83, "resize": 84,
    "itxfm_add": 85, "loop_filter_sb": 408, "cdef_dir": 412, "cdef_fb": 413, "wiener": 416, "sgr": 418,
}
COUNT = {"generate_grain_uv": 3, "fguv_32x32xn": 3, "intra_pred": 14, "cfl_ac": 3, "cfl_pred": 6, "mc": 10, "mc_scaled": 10,
         "mct": 10, "mct_scaled": 10, "w_mask": 3, "cdef_fb": 3, "wiener": 2, "sgr": 3}


def slot(family, i=0, j=0):
    if family == "itxfm_add":
        return BASE[family] + i * 17 + j
    if family == "loop_filter_sb":
        return BASE[family] + i * 2 + j
    return BASE[family] + i


class Table(util.Oracle):
    """An Oracle whose entries come out of a table of function pointers."""

    def __init__(self, ptrs, which):
        self.which = which
        self._cache = {}
        self._entry = lambda bpc, fam, i, j: ptrs[bpc][slot(fam.decode(), i, j)]


@pytest.fixture(scope="module")
def tables(ctx):
    if util.ref_lib() is None:
        pytest.skip("the table is compared with the table of the reference build (oracle/_ref)")
    lib = ctx.lib
    mine, ref = {}, {}
    rl = util.ref_lib()
    rl.dav1d_ref_dsp_context.restype = C.c_void_p
    rl.dav1d_ref_dsp_context.argtypes = [C.c_int]
    keep = []
    for bpc in (8, 10, 12):
        tab = (C.c_void_p * N_SLOTS)()
        rc = lib.dav1d_hip_dsp_init_8bpc(tab) if bpc == 8 else lib.dav1d_hip_dsp_init_16bpc(tab, bpc)
        assert rc == 0, rc
        keep.append(tab)
        mine[bpc] = [tab[k] for k in range(N_SLOTS)]
        r = (C.c_void_p * N_SLOTS).from_address(rl.dav1d_ref_dsp_context(bpc))
        ref[bpc] = [r[k] for k in range(N_SLOTS)]
    return Table(mine, "hip-table"), Table(ref, "ref-table"), mine, ref


def test_table_layout_is_the_reference_layout(tables):
    """Slot k of our table must be the member the reference build keeps at slot k."""
    _, _, mine, ref = tables
    o = util.Oracle("ref")
    for bpc in (8, 10, 12):
        for fam, base in BASE.items():
            if fam == "itxfm_add":
                idx = [(i, j) for i in range(19) for j in range(17)]
            elif fam == "loop_filter_sb":
                idx = [(i, j) for i in range(2) for j in range(2)]
            else:
                idx = [(i, 0) for i in range(COUNT.get(fam, 1))]
            for i, j in idx:
                want = o._entry(bpc, fam.encode(), i, j)
                assert ref[bpc][slot(fam, i, j)] == want, (bpc, fam, i, j)
        for k in range(N_SLOTS):
            assert bool(mine[bpc][k]) == bool(ref[bpc][k]), (bpc, k)


def both(tables, bpc, family, i, j, make_args, outputs):
    """Run the entry of both tables on identical copies of the arguments; `outputs` = indices of output arrays."""
    hip, ref = tables[0], tables[1]
    a, b = make_args(), make_args()
    ra = hip.call(bpc, family, i, j, *a)
    rb = ref.call(bpc, family, i, j, *b)
    for k in outputs:
        assert np.array_equal(a[k], b[k]), (family, i, j, bpc, k, np.argwhere(a[k] != b[k])[:4])
    return ra, rb


def frame(rng, bpc, h, w):
    return rng.integers(0, 1 << bpc, size=(h, w)).astype(util.pix_dtype(bpc))


class Ptr:
    """Pointer into the interior of an array (the array is kept alive by the caller)."""
    def __new__(cls, arr, *idx):
        return arr.ctypes.data + sum(int(i) * int(s) for i, s in zip(idx, arr.strides))


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_itx_through_table(tables, bpc):
    rng = np.random.default_rng(bpc)
    for tx, txtp in [(0, 0), (0, 16), (1, 5), (2, 11), (3, 9), (4, 0), (5, 3), (9, 0), (12, 0), (13, 15), (18, 0)]:
        coef, eob = util.gen_itx_coefs(rng, tx, txtp, bpc, util.subsh_max(tx) - 1)
        w, h = util.TX_W[tx], util.TX_H[tx]
        dst = frame(rng, bpc, h, w + 8)

        def args():
            return [dst.copy(), dst.strides[0], coef.copy(), eob]
        a, b = args(), args()
        tables[0].call(bpc, "itxfm_add", tx, txtp, *a)
        tables[1].call(bpc, "itxfm_add", tx, txtp, *b)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]), (tx, txtp)


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_mc_through_table(tables, bpc):
    rng = np.random.default_rng(20 + bpc)
    pd = util.pix_dtype(bpc)
    bps = pd().itemsize
    src = frame(rng, bpc, 300, 320)
    for f in (0, 4, 9):
        for w, h, mx, my in [(4, 4, 3, 0), (8, 16, 0, 7), (32, 8, 15, 1), (64, 64, 8, 8), (2, 4, 1, 1)]:
            sp = Ptr(src, 20, 24)
            both(tables, bpc, "mc", f, 0, lambda: [np.zeros((h, w + 3), pd), (w + 3) * bps, sp, src.strides[0], w, h, mx, my], [0])
            if w >= 4:
                both(tables, bpc, "mct", f, 0, lambda: [np.zeros(w * h, np.int16), sp, src.strides[0], w, h, mx, my], [0])
            dx, dy = int(rng.integers(1, 2049)), int(rng.integers(1, 2049))
            smx, smy = int(rng.integers(0, 1024)), int(rng.integers(0, 1024))
            both(tables, bpc, "mc_scaled", f, 0,
                 lambda: [np.zeros((h, w + 3), pd), (w + 3) * bps, sp, src.strides[0], w, h, smx, smy, dx, dy], [0])
            if w >= 4:
                both(tables, bpc, "mct_scaled", f, 0, lambda: [np.zeros(w * h, np.int16), sp, src.strides[0], w, h, smx, smy, dx, dy], [0])
    ib = 4 if bpc == 8 else 14 - bpc
    bias = 8192 if bpc > 8 else 0
    for w, h in [(4, 8), (16, 16), (128, 32)]:
        t1 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
        t2 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
        dst = frame(rng, bpc, h, w)
        m = rng.integers(0, 65, size=w * h).astype(np.uint8)
        both(tables, bpc, "avg", 0, 0, lambda: [dst.copy(), dst.strides[0], t1, t2, w, h], [0])
        both(tables, bpc, "w_avg", 0, 0, lambda: [dst.copy(), dst.strides[0], t1, t2, w, h, 5], [0])
        both(tables, bpc, "mask", 0, 0, lambda: [dst.copy(), dst.strides[0], t1, t2, w, h, m], [0])
        for ss in range(3):
            both(tables, bpc, "w_mask", ss, 0, lambda: [dst.copy(), dst.strides[0], t1, t2, w, h, np.zeros(w * h, np.uint8), 1], [0])
        if w <= 32:
            tmp = frame(rng, bpc, h, w)
            both(tables, bpc, "blend", 0, 0, lambda: [dst.copy(), dst.strides[0], tmp, w, h, m], [0])
            both(tables, bpc, "blend_v", 0, 0, lambda: [dst.copy(), dst.strides[0], tmp, w, h], [0])
        if h <= 32:
            tmp = frame(rng, bpc, h, w)
            both(tables, bpc, "blend_h", 0, 0, lambda: [dst.copy(), dst.strides[0], tmp, w, h], [0])
    # w_mask: the defined part of the mask output
    w, h = 16, 16
    t1 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
    t2 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
    for ss in range(3):
        n = (w >> (ss > 0)) * (h >> (ss == 2))
        outs = []
        for t in tables[:2]:
            m = np.zeros(w * h, np.uint8)
            t.call(bpc, "w_mask", ss, 0, np.zeros((h, w), pd), w * bps, t1, t2, w, h, m, 0)
            outs.append(m[:n].copy())
        assert np.array_equal(outs[0], outs[1])
    # warp, emu_edge, resize
    abcd = (rng.integers(0, 0x2000, size=4) - 0xa00).astype(np.int16)
    sp = Ptr(src, 40, 40)
    both(tables, bpc, "warp8x8", 0, 0, lambda: [np.zeros((8, 8), pd), 8 * bps, sp, src.strides[0], abcd, 1234, -600], [0])
    both(tables, bpc, "warp8x8t", 0, 0, lambda: [np.zeros((8, 8), np.int16), 8, sp, src.strides[0], abcd, -999, 77], [0])
    if bpc != 12:
        for x, y, bw, bh in [(-5, -7, 20, 19), (290, 310, 40, 30), (100, 100, 8, 8), (-60, 5, 30, 9)]:
            both(tables, bpc, "emu_edge", 0, 0, lambda: [bw, bh, 320, 300, x, y, np.zeros((bh, bw + 5), pd), (bw + 5) * bps,
                                                       src.ctypes.data, src.strides[0]], [6])
    src_w, dst_w = 200, 275
    dxr = ((src_w << 14) + (dst_w >> 1)) // dst_w
    both(tables, bpc, "resize", 0, 0, lambda: [np.zeros((12, dst_w + 1), pd), (dst_w + 1) * bps, src.ctypes.data, src.strides[0],
                                                dst_w, 12, src_w, dxr, 77], [0])


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_ipred_through_table(tables, bpc):
    """tests/checkasm/ipred.c:88-285: every mode, flag bits for the directional ones, cfl and palette."""
    rng = np.random.default_rng(30 + bpc)
    pd = util.pix_dtype(bpc)
    bps = pd().itemsize
    for mode in range(14):
        for w, h in [(4, 4), (8, 16), (16, 4), (32, 32), (64, 16), (16, 64)]:
            if mode == 13 and (w > 32 or h > 32):
                continue
            edge = frame(rng, bpc, 1, 257).ravel()
            tl = edge.ctypes.data + 128 * bps
            a = 0
            if 6 <= mode <= 8:
                a = (90 * (mode - 6) + int(rng.choice(np.arange(3, 90, 3)))) | (int(rng.integers(0, 4)) << 9)
                if mode == 7:
                    a = (90 + int(rng.choice(np.arange(3, 90, 3)))) | (int(rng.integers(0, 4)) << 9)
                if mode == 8:
                    a = (180 + int(rng.choice(np.arange(3, 90, 3)))) | (int(rng.integers(0, 4)) << 9)
                if mode == 6:
                    a = int(rng.choice(np.arange(3, 90, 3))) | (int(rng.integers(0, 4)) << 9)
            elif mode == 13:
                a = int(rng.integers(0, 5))
            mw, mh = int(rng.integers(1, w + 1)), int(rng.integers(1, h + 1))
            both(tables, bpc, "intra_pred", mode, 0, lambda: [np.zeros((h, w + 1), pd), (w + 1) * bps, tl, w, h, a, mw, mh], [0])
    for layout in range(3):
        ss_hor, ss_ver = layout < 2, layout == 0
        for cw, ch in [(4, 4), (8, 16), (16, 8), (32, 32)]:
            for w_pad, h_pad in [(0, 0), (cw // 4 - 1, 0), (0, ch // 4 - 1)]:
                luma = frame(rng, bpc, ch << ss_ver, (cw << ss_hor) + 2)
                both(tables, bpc, "cfl_ac", layout, 0, lambda: [np.zeros(cw * ch, np.int16), luma, luma.strides[0], w_pad, h_pad, cw, ch], [0])
    for mode in (0, 3, 4, 5):
        for w, h in [(4, 4), (8, 16), (32, 8)]:
            edge = frame(rng, bpc, 1, 257).ravel()
            tl = edge.ctypes.data + 128 * bps
            ac = rng.integers(-(1 << (bpc + 2)), 1 << (bpc + 2), size=w * h).astype(np.int16)
            alpha = int(rng.integers(-16, 17))
            both(tables, bpc, "cfl_pred", mode, 0, lambda: [np.zeros((h, w), pd), w * bps, tl, w, h, ac, alpha], [0])
    if bpc != 12:
        for w, h in [(4, 4), (8, 8), (64, 16)]:
            pal = frame(rng, bpc, 1, 8).ravel()
            idx = (rng.integers(0, 8, size=w * h // 2) | (rng.integers(0, 8, size=w * h // 2) << 4)).astype(np.uint8)
            both(tables, bpc, "pal_pred", 0, 0, lambda: [np.zeros((h, w), pd), w * bps, pal, idx, w, h], [0])


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_loop_filter_through_table(tables, bpc):
    rng = np.random.default_rng(40 + bpc)
    pd = util.pix_dtype(bpc)
    e, i = make_lut(int(rng.integers(0, 8)))
    lut = LutStruct()
    lut.e[:] = list(e); lut.i[:] = list(i)
    bd8 = bpc - 8
    changed = 0
    for chroma in (0, 1):
        for d in (0, 1):
            # gentle steps between 16x16 blocks and several noise levels: every filter width and the hev path fire
            base = np.kron(rng.integers(-12, 13, size=(10, 10)) << bd8, np.ones((16, 16), np.int64))
            amp = np.kron(rng.choice([0, 1, 2, 6], size=(10, 10)) << bd8, np.ones((16, 16), np.int64))
            plane = np.clip((1 << (bpc - 1)) + base + rng.integers(-1, 2, size=(160, 160)) * amp, 0, (1 << bpc) - 1).astype(pd)
            lvl = rng.integers(0, 64, size=(40, 40, 4)).astype(np.uint8)
            lvl[rng.random((40, 40)) < 0.2] = 0
            m = rng.integers(0, 1 << 32, size=3, dtype=np.uint64).astype(np.uint32)
            m[1] &= m[0]; m[2] &= m[1]        # wider filters only where the narrower bit is set, as lf_mask builds them
            if chroma:
                m[2] = 0
            m[0] &= np.uint32(0x00ffffff)      # 24 units: stay inside the 160-pixel plane with room for the widest filter
            m[1] &= m[0]; m[2] &= m[1]
            masks = np.ascontiguousarray(m)

            def args():
                p = plane.copy()
                return [p, Ptr(p, 16, 16), p.strides[0], masks, Ptr(lvl, 4, 4), 40, C.addressof(lut), 24]
            a, b = args(), args()
            tables[0].call(bpc, "loop_filter_sb", chroma, d, *a[1:])
            tables[1].call(bpc, "loop_filter_sb", chroma, d, *b[1:])
            assert np.array_equal(a[0], b[0]), (chroma, d, np.argwhere(a[0] != b[0])[:4])
            changed += int((a[0] != plane).sum())
    assert changed > 200, "the case must actually filter something"


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_cdef_through_table(tables, bpc):
    rng = np.random.default_rng(50 + bpc)
    pd = util.pix_dtype(bpc)
    bd8 = bpc - 8
    for it in range(6):
        blk = structured_plane(rng, (8, 8), bpc).astype(pd) if it & 1 else frame(rng, bpc, 8, 8)
        outs = []
        for t in tables[:2]:
            var = C.c_uint(0)
            a = [blk, blk.strides[0], C.addressof(var)]
            outs.append((t.call(bpc, "cdef_dir", 0, 0, *a), var.value))
        assert outs[0] == outs[1], outs
    for fb, (w, h) in enumerate([(8, 8), (4, 8), (4, 4)]):
        for edges in range(16):
            plane = structured_plane(rng, (16, 16), bpc).astype(pd)
            left = frame(rng, bpc, 8, 2)
            top = frame(rng, bpc, 2, 16)
            bot = frame(rng, bpc, 2, 16)
            pri = int(rng.integers(0, 16)) << bd8
            sec = (1 << int(rng.integers(0, 3))) << bd8 if rng.integers(0, 4) else 0
            if not pri and not sec:
                pri = 3 << bd8
            dirn, damping = int(rng.integers(0, 8)), int(rng.integers(3, 7)) + bd8

            def args():
                p = plane.copy()
                return [p, Ptr(p, 4, 4), p.strides[0], left, Ptr(top, 0, 4), Ptr(bot, 0, 4), pri, sec, dirn, damping, edges]
            a, b = args(), args()
            tables[0].call(bpc, "cdef_fb", fb, 0, *a[1:])
            tables[1].call(bpc, "cdef_fb", fb, 0, *b[1:])
            assert np.array_equal(a[0], b[0]), (fb, edges, pri, sec, dirn, damping)


class LrParams(C.Union):          # LooprestorationParams, reference src/looprestoration.h:49-55
    class _Sgr(C.Structure):
        _fields_ = [("s0", C.c_uint32), ("s1", C.c_uint32), ("w0", C.c_int16), ("w1", C.c_int16)]
    _fields_ = [("filter", (C.c_int16 * 8) * 2), ("sgr", _Sgr)]
    _align_ = 16


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_loop_restoration_through_table(tables, bpc):
    rng = np.random.default_rng(60 + bpc)
    pd = util.pix_dtype(bpc)
    sgr_params = util.ref_table("sgr_params", np.uint16).reshape(16, 2)
    for fam, idx in [("wiener", 0), ("wiener", 1), ("sgr", 0), ("sgr", 1), ("sgr", 2)]:
        for edges in (0, 5, 10, 15, 3, 12):
            w, h = int(rng.integers(1, 120)), int(rng.integers(1, 65))
            plane = frame(rng, bpc, 80, 160)
            left = frame(rng, bpc, 64, 4)
            lpf = frame(rng, bpc, 10, 160)
            prm = LrParams()
            if fam == "wiener":
                f = wiener_params(rng, bpc, idx == 1)
                for r in range(2):
                    for k in range(8):
                        prm.filter[r][k] = int(f[r][k])
            else:
                sets = [s for s in range(16) if (bool(sgr_params[s][0]) + 2 * bool(sgr_params[s][1]) - 1) == idx]
                s = int(rng.choice(sets))
                prm.sgr.s0, prm.sgr.s1 = int(sgr_params[s][0]), int(sgr_params[s][1])
                w0 = int(rng.integers(-96, 32))
                prm.sgr.w0 = w0 if idx != 1 else 0
                prm.sgr.w1 = (160 - int(rng.integers(0, 128)) - w0) if idx != 0 else 0
                if idx == 1:
                    prm.sgr.w1 = int(rng.integers(-32, 96))

            def args():
                p = plane.copy()
                return [p, Ptr(p, 8, 8), p.strides[0], left, Ptr(lpf, 0, 8), w, h, C.addressof(prm), edges]
            a, b = args(), args()
            tables[0].call(bpc, fam, idx, 0, *a[1:])
            tables[1].call(bpc, fam, idx, 0, *b[1:])
            assert np.array_equal(a[0][8:8 + h, 8:8 + w], b[0][8:8 + h, 8:8 + w]), (fam, idx, edges, w, h)
            # nothing outside the unit may change (the reference may write aligned garbage to the right: not compared)
            assert np.array_equal(a[0][:8], plane[:8]) and np.array_equal(a[0][:, :8], plane[:, :8])


@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_film_grain_through_table(tables, bpc):
    rng = np.random.default_rng(70 + bpc)
    pd = util.pix_dtype(bpc)
    ent = np.int8 if bpc == 8 else np.int16
    for variant in range(3):
        d = random_fg(rng, bpc, variant)
        luts = []
        for t in tables[:2]:
            gy = np.zeros((74, 82), ent)
            t.call(bpc, "generate_grain_y", 0, 0, gy, C.addressof(d))
            luts.append(gy)
        assert np.array_equal(luts[0][:73], luts[1][:73])
        gy = luts[1]
        for layout in range(3):
            for uv in range(2):
                outs = []
                for t in tables[:2]:
                    g = np.zeros((74, 82), ent)
                    t.call(bpc, "generate_grain_uv", layout, 0, g, gy, C.addressof(d), uv)
                    outs.append(g)
                assert np.array_equal(outs[0], outs[1]), ("gen_uv", layout, uv)
        scaling = rng.integers(0, 256, size=1 << bpc).astype(np.uint8)
        pw = 100
        for row_num, bh in [(0, 32), (3, 32), (5, 17)]:
            src = frame(rng, bpc, 32, 128)
            both(tables, bpc, "fgy_32x32xn", 0, 0,
                 lambda: [np.zeros((32, 128), pd), src, src.strides[0], C.addressof(d), pw, scaling, gy, bh, row_num], [0])
        guv = outs[1]
        for layout in range(3):
            sx, sy = layout < 2, layout == 0
            cpw = pw >> sx
            for row_num, lbh in [(0, 32), (2, 32), (4, 18)]:
                cbh = (lbh + sy) >> sy
                luma = frame(rng, bpc, 32, 128)
                csrc = frame(rng, bpc, 32, 128)
                for uv_pl in range(2):
                    for t_is_id in (0, 1):
                        g = np.zeros((74, 82), ent)
                        tables[1].call(bpc, "generate_grain_uv", layout, 0, g, gy, C.addressof(d), uv_pl)
                        both(tables, bpc, "fguv_32x32xn", layout, 0,
                             lambda: [np.zeros((32, 128), pd), csrc, csrc.strides[0], C.addressof(d), cpw, scaling, g, cbh, row_num,
                                      luma, luma.strides[0], uv_pl, t_is_id], [0])
