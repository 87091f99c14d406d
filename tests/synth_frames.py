"""Synthetic per-frame task lists for the reconstruction hot path.

No AV1 bitstreams are available, so bench.py and the frame-level tests build the
lists a pass-2 lister would emit for one inter frame: a block grid drawn from a size
mix, one or two motion vectors per block, dequantised coefficients in the valid
dynamic range with an eob consistent with the default scan order.  Everything is
derived from a seed (numpy PCG64), vectorised so that an 8K frame takes seconds.
"""
import os

import numpy as np

from dav1d_amd._lib import ITX_TASK, MC_TASK, COMP_TASK, LF_TASK, CDEF_TASK, LR_TASK, IPRED_TASK, FilmGrainData

HERE = os.path.dirname(os.path.abspath(__file__))

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
SQ_TX = {4: 0, 8: 1, 16: 2, 32: 3, 64: 4}

_scans = None


def scans():
    """Default scan order per tx size (AV1 spec tables; dumped once from the oracle build by
    tests/util.py into tests/data/scans.npz)."""
    global _scans
    if _scans is None:
        z = np.load(os.path.join(HERE, "data", "scans.npz"))
        _scans = [z["tx%d" % t].astype(np.int64) for t in range(19)]
    return _scans


def plane_geometry(w, h, bpc, layout=1):
    """(stride in pixels, padded rows) per plane; mirrors dav1d_hip_picture_alloc /
    reference src/picture.c:46-78."""
    hbd = bpc > 8
    aw, ah = (w + 127) & ~127, (h + 127) & ~127
    ss_ver = 1 if layout == 1 else 0
    ss_hor = 1 if layout != 3 else 0
    ys = aw << hbd
    uvs = ys >> ss_hor
    if not (ys & 1023):
        ys += 64
    if not (uvs & 1023):
        uvs += 64
    bps = 2 if hbd else 1
    return [(ys // bps, ah), (uvs // bps, ah >> ss_ver), (uvs // bps, ah >> ss_ver)]


def _dct_mat(n):
    i = np.arange(n)[:, None]
    j = np.arange(n)[None, :]
    m = np.cos(np.pi * (2 * j + 1) * i / (2.0 * n))
    m[0] *= np.sqrt(0.5)
    return m


_SCALE = [4.0, 4.0 * np.sqrt(0.5), 2.0, 2.0 * np.sqrt(0.5), 1.0, 0.5 * np.sqrt(0.5), 0.25, 0.125 * np.sqrt(0.5), 0.0625]


def gen_coefs(rng, tx, n, bpc, eob_class_p=(0.3, 0.4, 0.3)):
    """n slabs for tx size `tx` (2-D scan class): forward float DCT of a random residual in
    [-bitdepth_max, bitdepth_max] (like tests/checkasm/itx.c:185-242), rounded, everything after
    a per-block eob zeroed in scan order.  eob classes: dc-only / <= 1/4 of the slab / anywhere.
    Returns (coefs[n, sw*sh] in the reference's column-major slab layout, eob[n])."""
    w, h = TX_W[tx], TX_H[tx]
    sw, sh = min(w, 32), min(h, 32)
    bdmax = (1 << bpc) - 1
    cdt = np.int16 if bpc == 8 else np.int32
    scale = _SCALE[int(np.log2(w * h)) - 4]
    out = np.empty((n, sw * sh), cdt)
    ncoef = sw * sh
    cls = rng.choice(3, size=n, p=eob_class_p)
    eob = np.where(cls == 0, 0, np.where(cls == 1, rng.integers(1, max(2, ncoef // 4), size=n),
                                         rng.integers(1, ncoef, size=n))).astype(np.int64)
    scan = scans()[tx]
    inv = np.empty(ncoef, np.int64)
    inv[scan] = np.arange(ncoef)          # position of raster index rc in scan order
    mw, mh = _dct_mat(w)[:sw] * scale, _dct_mat(h)[:sh]
    step = max(1, (1 << 22) // (w * h))
    for s in range(0, n, step):
        e = min(n, s + step)
        res = rng.integers(-bdmax, bdmax + 1, size=(e - s, h, w)).astype(np.float32)
        t = np.einsum("nyx,kx->nyk", res, mw.astype(np.float32))      # rows -> [n, y, kx]
        c = np.einsum("ly,nyk->nkl", mh.astype(np.float32), t)        # cols -> [n, kx, ky] == slab[x*sh + y]
        c = np.floor(c + 0.5).reshape(e - s, ncoef)
        c[inv[None, :] > eob[s:e, None]] = 0
        out[s:e] = c.astype(cdt)
    return out, eob


class Frame:
    """Host-side task lists + coefficient arena of one synthetic inter frame."""
    pass


def make_frame(w, h, bpc, seed, mix=(0.20, 0.30, 0.30, 0.15, 0.05), compound_frac=0.25, n_refs=3,
               mv_range_px=64, edge_frac=0.05, region=64, alt_txtp_frac=0.3):
    """Block-size mix by area over 64x64 regions: (64, 32, 16, 8, 4)."""
    rng = np.random.default_rng(seed)
    geo = plane_geometry(w, h, bpc, 1)
    sizes = np.array([64, 32, 16, 8, 4])
    rx = np.arange(0, w, region)
    ry = np.arange(0, h, region)
    gx, gy = np.meshgrid(rx, ry)
    gx, gy = gx.ravel(), gy.ravel()
    rcls = rng.choice(5, size=len(gx), p=mix)

    mc_parts, comp_parts, itx_parts = [], [], []
    coef_parts = []
    cf_off = 0
    prep_off = 0
    n_samples = 0

    for ci, s in enumerate(sizes):
        sel = np.flatnonzero(rcls == ci)
        if not len(sel):
            continue
        k = region // s
        ox, oy = np.meshgrid(np.arange(k) * s, np.arange(k) * s)
        # raster order inside the region, regions in raster order (decode order of a 1-tile frame)
        bx = (gx[sel][:, None] + ox.ravel()[None, :]).ravel()
        by = (gy[sel][:, None] + oy.ravel()[None, :]).ravel()
        keep = (bx < w) & (by < h)
        bx, by = bx[keep], by[keep]
        nb = len(bx)
        comp = (rng.random(nb) < compound_frac) & (s >= 8)
        ref0 = rng.integers(0, n_refs, size=nb)
        ref1 = (ref0 + 1 + rng.integers(0, max(1, n_refs - 1), size=nb)) % n_refs
        mv = rng.integers(-mv_range_px * 8, mv_range_px * 8 + 1, size=(2, nb, 2))   # [ref, block, (y, x)] 1/8 pel
        # a band of blocks points far outside the picture to exercise edge emulation
        far = rng.random(nb) < edge_frac
        mv[:, far, :] *= 24
        filt = np.where(rng.random(nb) < 0.7, 0, rng.integers(0, 10, size=nb))

        for pl in range(3):
            ss = 1 if pl else 0
            if s == 4 and pl:
                # 4x4 luma: chroma is predicted / coded once per 8x8 (by its last 4x4)
                m = ((bx & 4) != 0) & ((by & 4) != 0)
                pbx, pby, pw = bx[m] >> 1 & ~3, by[m] >> 1 & ~3, 4
                sub = m
            else:
                pbx, pby, pw = bx >> ss, by >> ss, s >> ss
                sub = np.ones(nb, bool)
            npl = len(pbx)
            stride = geo[pl][0]
            dst_off = (pby * stride + pbx).astype(np.uint32)
            n_samples += npl * pw * pw
            for r in range(2):
                mvy, mvx = mv[r, sub, 0], mv[r, sub, 1]
                use = np.ones(npl, bool) if r == 0 else comp[sub]
                t = np.zeros(int(use.sum()), MC_TASK)
                if ss:
                    t["src_x"] = (pbx + (mvx >> 4))[use]
                    t["src_y"] = (pby + (mvy >> 4))[use]
                    t["mx"] = (mvx & 15)[use]
                    t["my"] = (mvy & 15)[use]
                else:
                    t["src_x"] = (pbx + (mvx >> 3))[use]
                    t["src_y"] = (pby + (mvy >> 3))[use]
                    t["mx"] = ((mvx & 7) << 1)[use]
                    t["my"] = ((mvy & 7) << 1)[use]
                t["w"] = t["h"] = pw
                t["filter_2d"] = filt[sub][use]
                t["plane"] = pl
                t["ref"] = (ref0 if r == 0 else ref1)[sub][use]
                c_sub = comp[sub][use]
                # single-reference blocks PUT straight into the picture; compound blocks PREP both
                t["kind"] = np.where(c_sub, 1, 0)
                nprep = int(c_sub.sum())
                offs = prep_off + np.arange(nprep, dtype=np.int64) * pw * pw
                d = dst_off[use].copy()
                d[c_sub] = offs
                t["dst_off"] = d
                if r == 0:
                    first_offs = offs
                    prep_off += nprep * pw * pw
                else:
                    ct = np.zeros(nprep, COMP_TASK)
                    ct["dst_off"] = dst_off[comp[sub]]
                    ct["tmp1_off"] = first_offs
                    ct["tmp2_off"] = offs
                    ct["w"] = ct["h"] = pw
                    ct["kind"] = 0
                    ct["plane"] = pl
                    comp_parts.append(ct)
                    prep_off += nprep * pw * pw
                mc_parts.append(t)
            # residual: one transform block per prediction block (64x64 chroma -> 32x32 etc.)
            tx = SQ_TX[pw]
            cf, eob = gen_coefs(rng, tx, npl, bpc)
            it = np.zeros(npl, ITX_TASK)
            it["dst_off"] = dst_off
            ncoef = cf.shape[1]
            it["cf_off"] = cf_off + np.arange(npl, dtype=np.int64) * ncoef
            it["eob"] = eob
            it["tx"] = tx
            txtp = np.zeros(npl, np.uint8)
            if pw <= 16:
                alt = (rng.random(npl) < alt_txtp_frac) & (eob > 0)
                txtp[alt] = rng.integers(1, 10, size=int(alt.sum()))      # 2-D classes only (scan order stays valid)
            elif pw == 32:
                alt = (rng.random(npl) < alt_txtp_frac * 0.3) & (eob > 0)
                txtp[alt] = 9
            it["txtp"] = txtp
            it["plane"] = pl
            itx_parts.append(it)
            coef_parts.append(cf.reshape(-1))
            cf_off += npl * ncoef

    f = Frame()
    f.w, f.h, f.bpc, f.n_refs = w, h, bpc, n_refs
    f.region, f.region_cls, f.region_sizes = region, rcls.reshape(len(ry), len(rx)), sizes
    f.mc = np.concatenate(mc_parts)
    f.comp = np.concatenate(comp_parts) if comp_parts else np.zeros(0, COMP_TASK)
    f.itx = np.concatenate(itx_parts)
    f.coef = np.concatenate(coef_parts)
    f.prep_elems = max(int(prep_off), 8)
    f.n_samples = int(n_samples)          # reconstructed samples (all planes)
    f.luma_pixels = w * h
    return f


def make_planes(rng, w, h, bpc, smooth=True, layout=1):
    """Padded random reference planes (3x3 box-smoothed noise), list of 3 arrays (1 for 4:0:0).  Each array is a
    (rows x cols) view into a buffer whose row stride equals the device picture's stride, so
    task offsets (y * stride + x) address host and device copies alike."""
    geo = plane_geometry(w, h, bpc, layout)
    pd = np.uint8 if bpc == 8 else np.uint16
    out = []
    for pl in range(1 if layout == 0 else 3):
        rows = geo[pl][1]
        cols = ((w + 127) & ~127) >> (1 if pl and layout != 3 else 0)
        a = rng.integers(0, 1 << bpc, size=(rows, cols), dtype=np.int32)
        if smooth:
            p = np.pad(a, 1, mode="edge")
            a = (p[:-2, :-2] + p[:-2, 1:-1] + p[:-2, 2:] + p[1:-1, :-2] + p[1:-1, 1:-1] + p[1:-1, 2:] +
                 p[2:, :-2] + p[2:, 1:-1] + p[2:, 2:] + 4) // 9
        base = np.zeros((rows, geo[pl][0]), pd)
        base[:, :cols] = a
        out.append(base[:, :cols])
    return out


def copy_planes(planes):
    """Deep copy that keeps each plane's row stride."""
    out = []
    for p in planes:
        base = np.zeros((p.shape[0], p.strides[0] // p.itemsize), p.dtype)
        base[:, :p.shape[1]] = p
        out.append(base[:, :p.shape[1]])
    return out


# ------------------------------------------------------------------ post-filter task lists for the same frame

class PostFilters:
    """Loop filter, CDEF, loop restoration and film grain inputs of one synthetic frame."""
    pass


def _pack_lines(bits, along_axis):
    """bits[class][.., ..] boolean per 4x4 unit -> uint32 masks per run of 32 units along `along_axis`."""
    out = []
    for b in bits:
        n = b.shape[along_axis]
        pad = (-n) % 32
        if pad:
            padw = [(0, 0), (0, 0)]
            padw[along_axis] = (0, pad)
            b = np.pad(b, padw)
        if along_axis == 0:
            r = b.reshape(b.shape[0] // 32, 32, b.shape[1])          # [line, u, x]
            m = (r.astype(np.uint64) << np.arange(32, dtype=np.uint64)[None, :, None]).sum(axis=1)   # [line, x]
        else:
            r = b.reshape(b.shape[0], b.shape[1] // 32, 32)          # [y, line, u]
            m = (r.astype(np.uint64) << np.arange(32, dtype=np.uint64)[None, None, :]).sum(axis=2)   # [y, line]
        out.append(m.astype(np.uint32))
    return out


def make_post_filters(frame, seed, layout=1):
    """Deblocking masks follow the frame's transform grid (one transform per block; chroma = half size, at least 4):
    an edge unit is filtered where a transform edge lies, with the width the smaller neighbour allows (16 / 8 / 4 on
    luma, 6 / 4 on chroma), levels 16..32; CDEF on every 8x8 with y strength 17, uv strength 5; Wiener on Y and
    SGR-mix on U / V in 64-pixel units and 64-row stripes (first stripe 8 rows short); film grain with 2 luma points,
    lag 3, overlap."""
    assert layout == 1
    rng = np.random.default_rng(seed)
    w, h, bpc = frame.w, frame.h, frame.bpc
    bd8 = bpc - 8
    geo = plane_geometry(w, h, bpc, layout)
    reg = frame.region
    size_of_region = frame.region_sizes[frame.region_cls]                 # luma transform size per 64x64 region
    p = PostFilters()

    # ---- loop filter
    w4, h4 = w // 4, h // 4
    b4_stride = (w4 + 31) & ~31
    p.b4_stride = b4_stride
    p.lvl = rng.integers(16, 33, size=((h4 + 31) & ~31, b4_stride, 4)).astype(np.uint8)
    p.lut_e, p.lut_i = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
    for level in range(64):            # dav1d_calc_eih with sharpness 0 (reference src/lf_mask.c:385-410)
        p.lut_i[level] = max(level, 1)
        p.lut_e[level] = 2 * (level + 2) + max(level, 1)
    tasks = []
    for pl in range(3):
        ss = 1 if pl else 0
        pw4, ph4 = w4 >> ss, h4 >> ss
        upr = (reg >> ss) // 4                                            # 4x4 units of this plane per region side
        tsz = np.maximum(size_of_region >> ss, 4)
        s_u = np.kron(tsz, np.ones((upr, upr), np.int64))[:ph4, :pw4]     # transform size at every unit
        yy, xx = np.mgrid[0:ph4, 0:pw4]
        for d in (0, 1):
            pos = (xx if d == 0 else yy) * 4
            here = s_u
            prev = np.roll(s_u, 1, axis=1 if d == 0 else 0)
            edge = (pos % here == 0) & (pos > 0)
            m = np.minimum(here, prev)
            if pl == 0:
                cls = [edge, edge & (m >= 8), edge & (m >= 16)]            # vmask[0] any width, [1] >= 8, [2] 16
            else:
                cls = [edge, edge & (m >= 8)]
            masks = _pack_lines(cls, 0 if d == 0 else 1)
            any_m = masks[0]
            li, lj = np.nonzero(any_m)                                    # d=0: (line = run of 32 rows, x4); d=1: (y4, line)
            t = np.zeros(len(li), LF_TASK)
            if d == 0:
                y4, x4 = li * 32, lj
            else:
                y4, x4 = li, lj * 32
            t["dst_off"] = (y4 * 4) * geo[pl][0] + x4 * 4
            t["lvl_off"] = y4 * b4_stride + x4
            for k in range(len(masks)):
                t["vmask"][:, k] = masks[k][li, lj]
            t["plane"], t["dir"] = pl, d
            t["lvl_comp"] = (0 if d == 0 else 1) if pl == 0 else 1 + pl
            tasks.append(t)
    p.lf = np.concatenate(tasks)

    # ---- cdef: every 8x8 unit, strengths as the frame header would give them (y 17 -> pri 4, sec 1; uv 5 -> pri 1, sec 1)
    bw, bh = w // 8, h // 8
    by, bx = np.mgrid[0:bh, 0:bw]
    c = np.zeros(bw * bh, CDEF_TASK)
    c["bx"], c["by"] = bx.ravel(), by.ravel()
    c["y_pri"], c["y_sec"], c["uv_pri"], c["uv_sec"] = 4 << bd8, 1 << bd8, 1 << bd8, 1 << bd8
    c["edges"] = ((bx > 0) * 1 + (bx < bw - 1) * 2 + (by > 0) * 4 + (by < bh - 1) * 8).ravel()
    p.cdef = c
    p.cdef_damping = 5 + bd8

    # ---- loop restoration: 64-pixel units; stripes of 64 luma rows, the first one 8 rows short (src/lr_apply_tmpl.c:50-51)
    lr = []
    for pl in range(3):
        ss = 1 if pl else 0
        pw, ph = w >> ss, h >> ss
        ys = [0]
        first = (64 - 8) >> ss
        y = first
        while y < ph:
            ys.append(y)
            y += 64 >> ss
        ys.append(ph)
        xs = np.arange(0, pw, 64)
        for si in range(len(ys) - 1):
            y0, y1 = ys[si], ys[si + 1]
            t = np.zeros(len(xs), LR_TASK)
            t["x"], t["y"] = xs, y0
            t["w"] = np.minimum(64, pw - xs)
            t["h"] = y1 - y0
            t["plane"] = pl
            t["edges"] = (xs > 0) * 1 + (xs + 64 < pw) * 2 + (4 if y0 > 0 else 0) + (8 if y1 < ph else 0)
            if pl == 0:
                t["type"] = 0                                            # 7-tap Wiener, taps in the checkasm ranges
                for d in range(2):
                    f0 = rng.integers(-5, 11, size=len(xs))
                    f1 = rng.integers(-23, 9, size=len(xs))
                    f2 = rng.integers(-17, 47, size=len(xs))
                    t["filter"][:, d, 0] = t["filter"][:, d, 6] = f0
                    t["filter"][:, d, 1] = t["filter"][:, d, 5] = f1
                    t["filter"][:, d, 2] = t["filter"][:, d, 4] = f2
                    centre = -(f0 + f1 + f2) * 2
                    t["filter"][:, d, 3] = centre + (128 if (d == 1 or bpc > 8) else 0)       # src/lr_apply_tmpl.c:55-66
            else:
                t["type"] = 4                                            # SGR mix, parameter set 0..9
                sgr = np.array([(140, 3236), (112, 2158), (93, 1618), (80, 1438), (70, 1295), (58, 1177), (47, 1079), (37, 996),
                                (30, 925), (25, 863)], np.int64)
                k = rng.integers(0, 10, size=len(xs))
                w0 = rng.integers(-96, 32, size=len(xs))
                w1 = 160 - rng.integers(0, 128, size=len(xs)) - w0
                t["filter"][:, 0, 0], t["filter"][:, 0, 1] = sgr[k, 0], sgr[k, 1]
                t["filter"][:, 0, 2], t["filter"][:, 0, 3] = w0, w1
            lr.append(t)
    p.lr = np.concatenate(lr)

    # ---- film grain
    d = FilmGrainData()
    d.seed = int(rng.integers(0, 1 << 16))
    d.num_y_points = 2
    d.y_points[0][0], d.y_points[0][1], d.y_points[1][0], d.y_points[1][1] = 16, 40, 235, 120
    for i in range(2):
        d.num_uv_points[i] = 2
        d.uv_points[i][0][0], d.uv_points[i][0][1], d.uv_points[i][1][0], d.uv_points[i][1][1] = 16, 30, 240, 90
        d.uv_mult[i], d.uv_luma_mult[i], d.uv_offset[i] = 64 + 8 * i, 32, 10 * (i + 1)
        for k in range(25):
            d.ar_coeffs_uv[i][k] = int(rng.integers(-32, 32))
    d.scaling_shift, d.ar_coeff_lag, d.ar_coeff_shift, d.grain_scale_shift = 10, 3, 7, 0
    for k in range(24):
        d.ar_coeffs_y[k] = int(rng.integers(-32, 32))
    d.overlap_flag, d.clip_to_restricted_range = 1, 0
    p.fg = d
    return p


# ------------------------------------------------------------------ an intra pass over part of the frame

class IntraPass:
    """Wavefront batches of intra blocks (prediction + residual) for a subset of the frame's 64x64 regions."""
    pass


def make_intra_pass(frame, seed, layout=1):
    """Every third region in x and y (1/9 of the frame, never two neighbours) is re-coded as intra with the block size it
    already has: the blocks of a region depend on their left / top / top-right neighbours, so block (i, j) of the region
    grid goes into wave j + 2 i (the classic 2:1 wavefront); everything outside the region is final by then.  One batch
    per wave: intra prediction of all its blocks in all planes, then their residuals.  Returns IntraPass with
    .batches = [(ipred_tasks, itx_tasks)], .coef (its own coefficient arena)."""
    assert layout == 1
    rng = np.random.default_rng(seed)
    w, h, bpc = frame.w, frame.h, frame.bpc
    geo = plane_geometry(w, h, bpc, layout)
    reg = frame.region
    nry, nrx = frame.region_cls.shape
    ry, rx = np.mgrid[0:nry, 0:nrx]
    sel = (ry % 3 == 1) & (rx % 3 == 1)
    waves = {}
    coef_parts, cf_off = [], 0
    for ci, s in enumerate(frame.region_sizes):
        m = sel & (frame.region_cls == ci)
        gy, gx = ry[m] * reg, rx[m] * reg
        if not len(gx):
            continue
        for pl in range(3):
            ss = 1 if pl else 0
            pw = max(int(s) >> ss, 4)                      # block = transform size in this plane
            k = (reg >> ss) // pw
            pwid, phei = w >> ss, h >> ss
            ii, jj = np.mgrid[0:k, 0:k]
            bx = ((gx >> ss)[:, None] + (jj.ravel() * pw)[None, :]).ravel()
            by = ((gy >> ss)[:, None] + (ii.ravel() * pw)[None, :]).ravel()
            wi = np.tile((jj + 2 * ii).ravel(), len(gx))
            jcol = np.tile(jj.ravel(), len(gx))
            irow = np.tile(ii.ravel(), len(gx))
            keep = (bx + pw <= pwid) & (by + pw <= phei)
            bx, by, wi, jcol, irow = bx[keep], by[keep], wi[keep], jcol[keep], irow[keep]
            n = len(bx)
            t = np.zeros(n, IPRED_TASK)
            t["dst_off"] = by * geo[pl][0] + bx
            t["x4"], t["y4"] = bx // 4, by // 4
            t["w4"], t["h4"] = pwid // 4, phei // 4
            t["tw"] = t["th"] = pw // 4
            mode = rng.integers(0, 14, size=n)
            if pw > 32:
                mode[mode == 13] = 12                      # filter-intra stops at 32x32
            t["mode"] = mode
            directional = (mode >= 1) & (mode <= 8)
            t["angle"] = np.where(directional, rng.integers(-3, 4, size=n), np.where(mode == 13, rng.integers(0, 5, size=n), 0))
            # edge availability as a decoder with 64x64 superblocks has it: the top-right neighbour is there for the region's top
            # row (the row above is final) and inside the region (an earlier wave), not right of the region below its top row
            # (that superblock comes later); the bottom-left one only for the region's left column above its last row
            flags = (bx > 0) * 1 + (by > 0) * 2 + ((irow == 0) | (jcol + 1 < k)) * 4 + ((jcol == 0) & (irow + 1 < k)) * 8 + 16 + rng.integers(0, 2, size=n) * 32
            t["flags"] = flags
            t["plane"], t["kind"] = pl, 0
            t["max_w"], t["max_h"] = pwid - bx, phei - by
            tx = SQ_TX[pw]
            cf, eob = gen_coefs(rng, tx, n, bpc)
            it = np.zeros(n, ITX_TASK)
            it["dst_off"] = t["dst_off"]
            it["cf_off"] = cf_off + np.arange(n, dtype=np.int64) * cf.shape[1]
            it["eob"], it["tx"], it["plane"] = eob, tx, pl
            coef_parts.append(cf.reshape(-1))
            cf_off += n * cf.shape[1]
            for d in np.unique(wi):
                q = wi == d
                a, b = waves.setdefault(int(d), ([], []))
                a.append(t[q]); b.append(it[q])
    p = IntraPass()
    p.batches = [(np.concatenate(waves[d][0]), np.concatenate(waves[d][1])) for d in sorted(waves)]
    p.coef = np.concatenate(coef_parts) if coef_parts else np.zeros(16, np.int16 if bpc == 8 else np.int32)
    p.n_blocks = int(sum(len(a) for a, _ in p.batches))
    p.n_samples = int(sum(int((a["tw"].astype(np.int64) * 4 * a["th"] * 4).sum()) for a, _ in p.batches))
    return p


def pack_frame_coefs(frame):
    """The frame's dense coefficient arena -> the sparse wire format (DAV1D_HIP_ITX_PACKED): per block only the eob + 1
    values in decode order.  Returns (itx tasks with PACKED set and cf_off into the packed arena, packed arena)."""
    t = frame.itx.copy()
    n = len(t)
    lens = t["eob"].astype(np.int64) + 1
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    packed = np.empty(int(lens.sum()), frame.coef.dtype)
    txtp = t["txtp"]
    cls = np.where(np.isin(txtp, (11, 13, 15)), 1, np.where(np.isin(txtp, (10, 12, 14)), 2, 0))
    for tx in np.unique(t["tx"]):
        sw, sh = min(TX_W[tx], 32), min(TX_H[tx], 32)
        nc = sw * sh
        i = np.arange(nc)
        order = [scans()[tx], i, (i & (sw - 1)) * sh + (i >> int(np.log2(sw)))]
        for c in range(3):
            sel = np.flatnonzero((t["tx"] == tx) & (cls == c))
            if not len(sel):
                continue
            dense = frame.coef[(t["cf_off"][sel].astype(np.int64)[:, None] + order[c][None, :])]       # [blocks, scan position]
            keep = i[None, :] <= t["eob"][sel].astype(np.int64)[:, None]
            vals = dense[keep]                                                                        # row-major: block by block
            dstpos = (offs[sel][:, None] + i[None, :])[keep]
            packed[dstpos] = vals
    t["cf_off"] = offs
    t["flags"] = 1
    return t, packed
