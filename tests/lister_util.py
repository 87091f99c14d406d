"""Frame-level parity plumbing: the reference's own pass 2 (oracle/ref_frame.c inside oracle/_ref) against
synthetic pass-1 output -> dav1d_hip_lister_* -> dav1d_hip_frame_* on the device (or the SIMT-emulated build).

The reference side is TEST INFRASTRUCTURE: it owns the hand-off arrays (allocated by the reference's
dav1d_decode_frame_init), the generator of tests/synth (dav1d_synth_frame) fills them, the reference reconstructs
from them on the CPU and the lister + kernels reconstruct from the very same arrays on the GPU."""
import ctypes as C

import numpy as np

import util
import synth_lib
from dav1d_amd import _lib, api


class RefFrameParams(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("layout", C.c_int), ("bpc", C.c_int), ("sb128", C.c_int), ("is_inter", C.c_int),
                ("n_tile_cols", C.c_int), ("n_tile_rows", C.c_int), ("col_start_sb", C.c_uint16 * 65), ("row_start_sb", C.c_uint16 * 65),
                ("intra_edge_filter", C.c_int), ("allow_screen_content_tools", C.c_int), ("switchable_comp_refs", C.c_int),
                ("ref_w", C.c_int * 7), ("ref_h", C.c_int * 7), ("ref_poc", C.c_int * 7), ("cur_poc", C.c_int),
                ("order_hint_n_bits", C.c_int), ("gmv_type", C.c_int * 7), ("gmv_matrix", (C.c_int32 * 6) * 7),
                ("lf_level_y", C.c_int * 2), ("lf_level_u", C.c_int), ("lf_level_v", C.c_int), ("lf_sharpness", C.c_int),
                ("lf_mode_ref_delta_enabled", C.c_int), ("lf_ref_delta", C.c_int * 8), ("lf_mode_delta", C.c_int * 2),
                ("cdef_enabled", C.c_int), ("cdef_damping", C.c_int), ("cdef_n_bits", C.c_int), ("cdef_y_strength", C.c_int * 8),
                ("cdef_uv_strength", C.c_int * 8), ("lr_type", C.c_int * 3), ("lr_unit_size", C.c_int * 2), ("sr_w", C.c_int), ("delta_lf", C.c_int),
                ("seg_enabled", C.c_int), ("seg_delta_lf", (C.c_int * 4) * 8), ("seg_lossless", C.c_int * 8)]


def ref_lib():
    lib = util.ref_lib()
    if lib is None:
        return None
    lib.dav1d_ref_frame_create.restype = C.c_void_p
    lib.dav1d_ref_frame_create.argtypes = [C.POINTER(RefFrameParams)]
    lib.dav1d_ref_frame_ptr.restype = C.c_void_p
    lib.dav1d_ref_frame_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]
    lib.dav1d_ref_frame_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.dav1d_ref_frame_recon.argtypes = [C.c_void_p]
    if hasattr(lib, "dav1d_ref_frame_recon_mt"):
        lib.dav1d_ref_frame_recon_mt.argtypes = [C.c_void_p, C.c_int]
    lib.dav1d_ref_frame_destroy.argtypes = [C.c_void_p]
    lib.dav1d_ref_layouts.argtypes = [C.POINTER(C.c_int)]
    lib.dav1d_ref_frame_build_filter_inputs.argtypes = [C.c_void_p, C.c_uint]
    lib.dav1d_ref_frame_filter.argtypes = [C.c_void_p]
    return lib


def uniform_tiles(n_sb, n_tiles):
    """start of every tile in superblocks, uniform spacing (reference src/obu.c:637-644)"""
    n_tiles = max(1, min(n_tiles, n_sb))
    size = (n_sb + n_tiles - 1) // n_tiles
    starts = list(range(0, n_sb, size))
    return starts + [n_sb]


class RefFrame:
    """One synthetic frame inside a real Dav1dFrameContext of the reference build."""

    def __init__(self, w, h, layout, bpc, is_inter=True, sb128=True, tile_cols=1, tile_rows=1, ref_sizes=None, gmv=None,
                 intra_edge_filter=1, screen_content=0, order_hint_bits=5, filters=None, sr_w=0, delta_lf=0, segments=None):
        self.lib = ref_lib()
        assert self.lib is not None, "the reference build oracle/_ref is required"
        p = RefFrameParams()
        p.w, p.h, p.layout, p.bpc, p.sb128, p.is_inter = w, h, layout, bpc, int(sb128), int(is_inter)
        sb = 128 if sb128 else 64
        cs = uniform_tiles((w + sb - 1) // sb, tile_cols)
        rs = uniform_tiles((h + sb - 1) // sb, tile_rows)
        p.n_tile_cols, p.n_tile_rows = len(cs) - 1, len(rs) - 1
        for i, v in enumerate(cs):
            p.col_start_sb[i] = v
        for i, v in enumerate(rs):
            p.row_start_sb[i] = v
        p.intra_edge_filter = intra_edge_filter
        p.allow_screen_content_tools = screen_content
        p.switchable_comp_refs = int(is_inter)
        p.order_hint_n_bits = order_hint_bits
        p.cur_poc = 8
        for i in range(7):
            rw, rh = (ref_sizes[i] if ref_sizes else (w, h))
            p.ref_w[i], p.ref_h[i] = rw, rh
            p.ref_poc[i] = [7, 6, 4, 2, 9, 10, 12][i]
            if gmv and gmv[i] is not None:
                p.gmv_type[i] = gmv[i][0]
                for k in range(6):
                    p.gmv_matrix[i][k] = gmv[i][1][k]
        if filters:
            # filters: dict(lf=(y0, y1, u, v, sharpness, deltas?), cdef=(damping, n_bits, y_strengths, uv_strengths), lr=(types[3], unit_log2[2]))
            if "lf" in filters:
                y0, y1, u, v, sharp, deltas = filters["lf"]
                p.lf_level_y[0], p.lf_level_y[1], p.lf_level_u, p.lf_level_v, p.lf_sharpness = y0, y1, u, v, sharp
                if deltas:
                    p.lf_mode_ref_delta_enabled = 1
                    for i, d in enumerate((1, 0, -1, 0, -1, 2, -2, 1)):
                        p.lf_ref_delta[i] = d
                    p.lf_mode_delta[0], p.lf_mode_delta[1] = 1, -1
            if "cdef" in filters:
                damping, n_bits, ys, uvs = filters["cdef"]
                p.cdef_enabled, p.cdef_damping, p.cdef_n_bits = 1, damping, n_bits
                for i in range(1 << n_bits):
                    p.cdef_y_strength[i], p.cdef_uv_strength[i] = ys[i], uvs[i]
            if "lr" in filters:
                types, units = filters["lr"]
                for i in range(3):
                    p.lr_type[i] = types[i]
                p.lr_unit_size[0], p.lr_unit_size[1] = units
        p.delta_lf = delta_lf
        if segments:
            # segments: dict(delta_lf=[[y_v, y_h, u, v]] * n, lossless=[0 / 1] * n): segmentation with per-segment level deltas
            # (Dav1dSegmentationData.delta_lf_*) and lossless segments (frame_hdr->segmentation.lossless)
            p.seg_enabled = 1
            for s_, dl in enumerate(segments.get("delta_lf", [])):
                for k in range(4):
                    p.seg_delta_lf[s_][k] = dl[k]
            for s_, v in enumerate(segments.get("lossless", [])):
                p.seg_lossless[s_] = int(v)
        p.sr_w = sr_w if sr_w and sr_w != w else 0
        self.sr_w = p.sr_w
        self.filters = filters
        self.p = p
        self.h = self.lib.dav1d_ref_frame_create(C.byref(p))
        assert self.h, "dav1d_ref_frame_create failed"
        geo = (C.c_int64 * 24)()
        self.lib.dav1d_ref_frame_geometry(self.h, geo)
        self.sr_stride = (int(geo[21]), int(geo[22]))
        self.b4_stride, self.bw, self.bh, self.sb128w, self.sbh = [int(geo[i]) for i in range(5)]
        self.cur_stride = (int(geo[5]), int(geo[6]))
        self.ref_stride = [(int(geo[7 + 2 * i]), int(geo[8 + 2 * i])) for i in range(7)]
        self.w, self.ht, self.layout, self.bpc, self.is_inter = w, h, layout, bpc, is_inter
        self.cols, self.rows = cs, rs

    def ptr(self, name):
        n = C.c_size_t()
        p = self.lib.dav1d_ref_frame_ptr(self.h, name.encode(), C.byref(n))
        return p, n.value

    def array(self, name, dtype):
        p, n = self.ptr(name)
        if not p:
            return None
        dt = np.dtype(dtype)
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p)).view(dt)

    def plane(self, slot, pl):
        """(rows x stride) view of a plane of picture slot 0 (current) or 1 + i (reference i), in pixels"""
        a = self.array("pic%d_%d" % (slot, pl), np.uint8 if self.bpc == 8 else np.uint16)
        stride = (self.cur_stride if slot == 0 else self.sr_stride if slot == 8 else self.ref_stride[slot - 1])[1 if pl else 0]
        spx = stride // a.itemsize
        return a[:(len(a) // spx) * spx].reshape(-1, spx)

    def desc(self):
        d = _lib.FrameDesc()
        p = self.p
        d.w, d.h, d.layout, d.bpc, d.sb128 = p.w, p.h, p.layout, p.bpc, p.sb128
        d.intra_edge_filter, d.is_inter = p.intra_edge_filter, p.is_inter
        d.n_tile_cols, d.n_tile_rows = p.n_tile_cols, p.n_tile_rows
        for i in range(65):
            d.col_start_sb[i] = p.col_start_sb[i]
            d.row_start_sb[i] = p.row_start_sb[i]
        d.b4_stride = self.b4_stride
        d.b = self.ptr("b")[0]
        d.cbi = self.ptr("cbi")[0]
        d.tile_start_off = self.ptr("tile_start_off")[0]
        d.pal = self.ptr("pal")[0]
        svc = self.array("svc", np.int32).reshape(7, 2, 2)
        gwa = self.array("gmv_warp_allowed", np.uint8)
        jw = self.array("jnt_weights", np.uint8).reshape(7, 7)
        gmv = self.array("gmv", np.uint8).reshape(7, -1)
        for i in range(7):
            for k in range(2):
                d.svc[i][k][0], d.svc[i][k][1] = int(svc[i, k, 0]), int(svc[i, k, 1])
            d.ref_w[i], d.ref_h[i] = p.ref_w[i], p.ref_h[i]
            d.gmv_warp_allowed[i] = int(gwa[i])
            for j in range(7):
                d.jnt_weights[i][j] = int(jw[i, j])
            C.memmove(C.addressof(d.gmv[i]), gmv[i].ctypes.data, C.sizeof(_lib.WarpParams))
        d.cf_align64 = 1                     # the oracle build is an x86-64 build (oracle/ref_config.h)
        for s_ in range(8):
            d.lossless[s_] = p.seg_lossless[s_]
        return d

    def recon(self, threads=1):
        """pass 2 of the frame: dav1d_decode_tile_sbrow over every tile; threads > 1: one worker per tile in flight"""
        rc = self.lib.dav1d_ref_frame_recon(self.h) if threads <= 1 else self.lib.dav1d_ref_frame_recon_mt(self.h, threads)
        assert rc == 0, "reference pass 2 failed"

    def build_filter_inputs(self, seed):
        """the deblocking masks / levels / cdef indices the reference's pass 1 would have built for the blocks in place, and
        random restoration units"""
        rc = self.lib.dav1d_ref_frame_build_filter_inputs(self.h, seed)
        assert rc == 0
        rng = np.random.default_rng(seed)
        lrm = self.array("lr_mask", np.int8)
        if lrm is not None and len(lrm):
            u = lrm.reshape(-1, 3, 4, 9)           # Av1Restoration.lr[plane][unit]{type, filter_h[3], filter_v[3], sgr_weights[2]}
            sgr = np.array([(140, 3236), (112, 2158), (93, 1618), (80, 1438), (70, 1295), (58, 1177), (47, 1079), (37, 996), (30, 925),
                            (25, 863), (0, 2589), (0, 1618), (0, 1177), (0, 925), (56, 0), (22, 0)])
            for pl in range(3):
                ft = self.p.lr_type[pl]            # 0 none, 1 switchable, 2 Wiener, 3 self-guided
                n = u.shape[0] * 4
                kind = rng.integers(0, 3, size=n) if ft == 1 else np.where(rng.random(n) < 0.8, ft - 1, 0) if ft else np.zeros(n, np.int64)
                idx = rng.integers(0, 16, size=n)
                typ = np.where(kind == 0, 0, np.where(kind == 1, 2, 3 + idx))
                uu = np.zeros((n, 9), np.int64)
                uu[:, 0] = typ
                for d in (1, 4):                   # filter_h, filter_v: the ranges of read_restoration_info(), src/decode.c:2531-2546
                    uu[:, d + 0] = 0 if pl else rng.integers(-5, 11, size=n)
                    uu[:, d + 1] = rng.integers(-23, 9, size=n)
                    uu[:, d + 2] = rng.integers(-17, 47, size=n)
                w0 = rng.integers(-96, 32, size=n)
                w1 = rng.integers(-32, 96, size=n)
                uu[:, 7] = np.where(sgr[idx, 0] != 0, w0, 0)
                uu[:, 8] = np.where(sgr[idx, 1] != 0, w1, 95)
                u[:, pl] = uu.reshape(-1, 4, 9).astype(np.int8)

    def filter(self):
        assert self.lib.dav1d_ref_frame_filter(self.h) == 0
        self.filtered_is_upscaled = bool(self.sr_w)      # the final picture is sr_cur (slot 8) then

    def filter_desc(self):
        fd = _lib.FilterDesc()
        p = self.p
        fd.lf_level_y[0], fd.lf_level_y[1], fd.lf_level_u, fd.lf_level_v = p.lf_level_y[0], p.lf_level_y[1], p.lf_level_u, p.lf_level_v
        fd.lf_mask = self.ptr("lf_mask")[0]
        fd.tx_lpf_right_edge[0] = self.ptr("tx_lpf_right_edge0")[0]
        fd.tx_lpf_right_edge[1] = self.ptr("tx_lpf_right_edge1")[0]
        lay = (C.c_int * 64)()
        self.lib.dav1d_ref_layouts(lay)
        v = list(lay)
        v = v[:v.index(-1)]
        a_sz, a_y, a_uv = v[-6], v[-5], v[-4]
        a = self.ptr("a")[0]
        fd.a_tx_lpf_y, fd.a_tx_lpf_uv, fd.a_stride = a + a_y, a + a_uv, a_sz
        fd.cdef_enabled, fd.cdef_damping = p.cdef_enabled, p.cdef_damping
        for i in range(8):
            fd.cdef_y_strength[i], fd.cdef_uv_strength[i] = p.cdef_y_strength[i], p.cdef_uv_strength[i]
        for i in range(3):
            fd.lr_type[i] = p.lr_type[i]
        fd.lr_unit_size[0], fd.lr_unit_size[1] = p.lr_unit_size[0], p.lr_unit_size[1]
        fd.lr_mask = self.ptr("lr_mask")[0]
        fd.sr_w = self.sr_w
        return fd

    def destroy(self):
        if self.h:
            self.lib.dav1d_ref_frame_destroy(self.h)
            self.h = None


def default_synth(seed, **kw):
    sp = synth_lib.SynthParams()
    sp.seed = seed
    sp.intra_pct, sp.skip_pct = 15, 20
    sp.compound_pct, sp.masked_compound = 30, 1
    sp.global_pct = 5
    sp.interintra_pct, sp.obmc_pct, sp.warp_pct = 15, 20, 10
    sp.cfl_pct, sp.palette, sp.filter_intra_pct = 30, 0, 25
    sp.tx_split_pct, sp.alt_txtp_pct, sp.eob_none_pct = 30, 40, 10
    sp.mv_range, sp.far_mv_pct, sp.n_refs = 256, 4, 7
    for i, v in enumerate((95, 80, 60, 45, 30)):
        sp.split_pct[i] = v
    sp.rect_pct = 50
    sp.fixed_bl = -1
    sp.cf_align64 = 1
    for k, v in kw.items():
        if k == "split_pct":
            for i, x in enumerate(v):
                sp.split_pct[i] = x
        else:
            setattr(sp, k, v)
    return sp


def synth(ctx, rf, sp):
    """run the product's generator on the reference-owned arrays"""
    d = rf.desc()
    cf, cf_bytes = rf.ptr("cf")
    _, cbi_bytes = rf.ptr("cbi")
    pal_idx, pal_idx_bytes = rf.ptr("pal_idx")
    C.memset(cf, 0, cf_bytes)
    rc = synth_lib.synth_frame(d, sp, cf, cf_bytes, cbi_bytes // 2, pal_idx, pal_idx_bytes)
    assert rc == 0, "dav1d_synth_frame: %d" % rc
    # the reference's itxfm_add consumes (zeroes) the coefficients: keep what pass 1 "produced" for the device side
    rf.cf_copy = rf.array("cf", np.uint8).copy()
    return d


def fill_pictures(rf, seed):
    """random (smoothed) reference pictures, random noise in the current picture's visible area is NOT needed: every
    pixel of it is written by the reconstruction; padding starts as zero on both sides"""
    rng = np.random.default_rng(seed)
    n_pl = 1 if rf.layout == 0 else 3
    if rf.is_inter:
        for i in range(7):
            for pl in range(n_pl):
                a = rf.plane(1 + i, pl)
                v = rng.integers(0, 1 << rf.bpc, size=a.shape, dtype=np.int32)
                p = np.pad(v, 1, mode="edge")
                v = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] + 4 * p[1:-1, 1:-1] + 4) // 8
                a[...] = v.astype(a.dtype)


def run_hip(ctx, rf, d, threads=1, with_filters=False, own_masks=False, packed=False, interleave=False):
    """lister -> frame API -> kernels; returns the reconstructed planes (visible area) and the lister handle stats.
    packed: the lister gathers the coefficients that exist out of the host arena (Dav1dHipFrameDesc.cf) into the frame's own
    coefficient arena, zeroing them in place; no dense arena goes to the device."""
    n_pl = 1 if rf.layout == 0 else 3
    cur = ctx.picture(rf.w, rf.ht, rf.layout, rf.bpc)
    refs = []
    if rf.is_inter:
        for i in range(7):
            r = ctx.picture(rf.p.ref_w[i], rf.p.ref_h[i], rf.layout, rf.bpc)
            for pl in range(n_pl):
                src = rf.plane(1 + i, pl)
                rows, cols = r.padded_shape(pl)
                r.upload(pl, np.ascontiguousarray(src[:rows, :cols]))
            refs.append(r)
    frame = ctx.frame(cur, refs)
    lh = C.c_void_p()
    host_cf = None
    if packed:
        host_cf = rf.cf_copy.copy()
        d.cf = host_cf.ctypes.data
    rc = ctx.lib.dav1d_hip_lister_create(C.byref(lh), C.byref(d), frame.h)
    d.cf = None
    assert rc == 0, "lister_create: %d" % rc
    jobs = [(tr, tc) for tr in range(d.n_tile_rows) for tc in range(d.n_tile_cols)]

    def one_tile(job):
        tr, tc = job
        for sby in range(d.row_start_sb[tr], d.row_start_sb[tr + 1]):
            rc2 = ctx.lib.dav1d_hip_lister_tile_sbrow(lh, tr, tc, sby)
            assert rc2 == 0, "lister_tile_sbrow(%d, %d, %d): %d" % (tr, tc, sby, rc2)
    if interleave:
        # tile-sbrows of different tiles come in any order (dav1d's task scheduling): the k-th superblock row of every tile, lowest
        # tile row first, then the (k + 1)-th
        k, left = 0, True
        while left:
            left = False
            for tr, tc in reversed(jobs):
                sby = d.row_start_sb[tr] + k
                if sby < d.row_start_sb[tr + 1]:
                    left = True
                    rc2 = ctx.lib.dav1d_hip_lister_tile_sbrow(lh, tr, tc, sby)
                    assert rc2 == 0, "lister_tile_sbrow(%d, %d, %d): %d" % (tr, tc, sby, rc2)
            k += 1
    elif threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one_tile, jobs))
    else:
        for j in jobs:
            one_tile(j)
    prep_elems = ctx.lib.dav1d_hip_lister_prep_elems(lh)
    mask_bytes = ctx.lib.dav1d_hip_lister_mask_bytes(lh)
    steps = ctx.lib.dav1d_hip_lister_steps(lh)
    # arenas: coefficients = the reference's cf array verbatim; aux = its packed palette indices
    coef = None if packed else ctx.buffer_from(rf.cf_copy)
    prep = ctx.buffer(prep_elems * 2 + 64)
    mask = ctx.buffer(mask_bytes + 64)
    nb = C.c_size_t()
    blob = ctx.lib.dav1d_hip_lister_const_masks(C.byref(nb))
    mask.upload(np.ctypeslib.as_array((C.c_uint8 * nb.value).from_address(blob)))
    aux = None
    pi = rf.array("pal_idx", np.uint8)
    if pi is not None and len(pi):
        aux = ctx.buffer_from(pi)
        frame.submit_intra_step(0, np.zeros(0, api.IPRED_TASK), np.zeros(0, api.ITX_TASK), aux)
    lvl = None
    if with_filters:
        fd = rf.filter_desc()
        sbh = rf.sbh
        keep = None
        if own_masks:
            # masks, noskip_mask, level cache and tile-edge contexts built by the product (dav1d_hip_lf_rects + _lf_masks_build)
            # instead of taken from the reference's pass 1; cdef_idx comes from the bitstream, so it is carried over
            lflvl = rf.array("lflvl", np.uint8)
            rects_p, n = C.c_void_p(), C.c_size_t()
            sbt = rf.array("sb_lflvl", np.uint8)            # delta_lf: one level table per superblock
            assert ctx.lib.dav1d_hip_lf_rects_sb(C.byref(d), lflvl.ctypes.data, sbt.ctypes.data if sbt is not None and len(sbt) else None,
                                                 C.byref(rects_p), C.byref(n)) == 0
            bw, bh = ((rf.w + 7) >> 3) << 1, ((rf.ht + 7) >> 3) << 1
            sb128w, sb128h, align_h = (bw + 31) >> 5, (bh + 31) >> 5, (bh + 31) & ~31
            ref_masks = rf.array("lf_mask", np.uint8).reshape(sb128w * sb128h, 1348)
            masks = np.zeros((sb128w * sb128h, 1348), np.uint8)
            lvl = ctx.buffer(sb128h * 32 * d.b4_stride * 4 + 64)
            lvl.zero()
            r_y, r_uv = np.zeros(align_h * d.n_tile_cols, np.uint8), np.zeros(align_h * d.n_tile_cols, np.uint8)
            a_y, a_uv = np.zeros(d.n_tile_rows * sb128w * 32, np.uint8), np.zeros(d.n_tile_rows * sb128w * 32, np.uint8)
            right = (C.c_void_p * 2)(r_y.ctypes.data, r_uv.ctypes.data)
            rc4 = ctx.lib.dav1d_hip_lf_masks_build(ctx.h, C.byref(d), rects_p, n.value, masks.ctypes.data, lvl.ptr, right,
                                                   a_y.ctypes.data, a_uv.ctypes.data)
            ctx.lib.dav1d_hip_lf_rects_free(rects_p)
            assert rc4 == 0, rc4
            masks[:, 1280:1284] = ref_masks[:, 1280:1284]
            fd.lf_mask = masks.ctypes.data
            fd.tx_lpf_right_edge[0], fd.tx_lpf_right_edge[1] = r_y.ctypes.data, r_uv.ctypes.data
            fd.a_tx_lpf_y, fd.a_tx_lpf_uv, fd.a_stride = a_y.ctypes.data, a_uv.ctypes.data, 32
            keep = (masks, r_y, r_uv, a_y, a_uv)
        for sby in range(sbh):
            rc3 = ctx.lib.dav1d_hip_lister_filter_sbrow(lh, C.byref(fd), sby)
            assert rc3 == 0, "lister_filter_sbrow(%d): %d" % (sby, rc3)
        if not own_masks:
            lvl = ctx.buffer_from(rf.array("lf_level", np.uint8))
        lut = rf.array("lim_lut", np.uint8)          # Av1FilterLUT: e[64], i[64], sharp[2]
        frame.set_filters(lvl, rf.b4_stride, lut[0:64], lut[64:128], rf.p.cdef_damping + rf.bpc - 8)
        if rf.sr_w:
            assert ctx.lib.dav1d_hip_frame_set_super_res(frame.h, rf.sr_w) == 0
    filtered = frame.end(coef, prep, mask)
    if with_filters:
        fpic = api.DevicePicture.view(ctx, filtered, rf.sr_w or rf.w, rf.ht, rf.layout, rf.bpc)
        out = [fpic.download(pl) for pl in range(n_pl)]
    else:
        out = [cur.download(pl) for pl in range(n_pl)]
    coef_after = host_cf if packed else coef.download(np.uint8)
    ctx.lib.dav1d_hip_lister_destroy(lh)
    frame.destroy()
    for b in (coef, prep, mask, aux, lvl):
        if b is not None:
            b.free()
    cur.free()
    for r in refs:
        r.free()
    return out, dict(prep_elems=prep_elems, mask_bytes=mask_bytes, steps=steps, coef_after=coef_after)


def compare(rf, got):
    """visible area of every plane against the reference's reconstruction"""
    n_pl = 1 if rf.layout == 0 else 3
    ss_hor = 1 if rf.layout in (1, 2) else 0
    ss_ver = 1 if rf.layout == 1 else 0
    bad = []
    slot = 8 if getattr(rf, "sr_w", 0) and getattr(rf, "filtered_is_upscaled", False) else 0
    fw = rf.sr_w if slot == 8 else rf.w
    for pl in range(n_pl):
        w = fw if not pl else (fw + ss_hor) >> ss_hor
        h = rf.ht if not pl else (rf.ht + ss_ver) >> ss_ver
        want = rf.plane(slot, pl)[:h, :w]
        have = got[pl][:h, :w]
        if not np.array_equal(want, have):
            yy, xx = np.nonzero(want != have)
            bad.append((pl, len(yy), int(yy[0]), int(xx[0]), int(want[yy[0], xx[0]]), int(have[yy[0], xx[0]])))
    return bad


def check_handoff_against_reference(ho, planes, ref_pics, is_inter=True):
    """The parity gate of the end-to-end route (tests/e2e.py): the hand-off arrays `ho` go into a real Dav1dFrameContext,
    the reference's OWN pass 2 (dav1d_decode_tile_sbrow, oracle/ref_frame.c) reconstructs the frame on the CPU, and the
    planes the device produced must equal it.  Returns a description, raises AssertionError on a difference."""
    import time
    if ref_lib() is None:
        return "skipped (no reference build)"
    rf = RefFrame(ho.w, ho.h, ho.layout, ho.bpc, is_inter=is_inter, sb128=bool(ho.sb128), tile_cols=ho.desc.n_tile_cols,
                  tile_rows=ho.desc.n_tile_rows)
    try:
        for name, src in (("b", ho.b), ("cbi", ho.cbi.view(np.uint8)), ("cf", ho.cf)):
            dst = rf.array(name, np.uint8)
            assert len(dst) == len(src), (name, len(dst), len(src))
            dst[:] = src
        assert np.array_equal(rf.array("tile_start_off", np.uint32)[:len(ho.tile_start_off)], ho.tile_start_off)
        if is_inter:
            for i in range(7):
                for pl in range(3):
                    hostp = ref_pics[i % len(ref_pics)].download(pl)
                    dstp = rf.plane(1 + i, pl)
                    dstp[:hostp.shape[0], :hostp.shape[1]] = hostp
        t0 = time.perf_counter()
        rf.recon()
        t_ref = time.perf_counter() - t0
        bad = compare(rf, planes)
        assert not bad, bad
        return "bit-exact vs the reference's own pass 2 (dav1d_decode_tile_sbrow, 1 thread: %.2f s for this frame)" % t_ref
    finally:
        rf.destroy()


def reference_pass2_rate(lib_ctx, w, h, bpc, tile_cols=16, tile_rows=8, threads=(32, 64, 128), seed=0xCB0):
    """The CPU peer of the end-to-end route: the reference's OWN pass 2 (dav1d_decode_tile_sbrow on a real Dav1dFrameContext,
    C DSP functions of this build) on the same kind of synthetic inter frame, split into tile_cols x tile_rows tiles and run
    by a pool of workers, one tile each at a time — the split dav1d's frame threading makes in pass 2.  Returns
    {threads: Mpixels/s}.  lib_ctx: anything with .lib = the product library (for the frame generator)."""
    import time
    import e2e
    if ref_lib() is None:
        return None
    rf = RefFrame(w, h, 1, bpc, is_inter=True, sb128=True, tile_cols=tile_cols, tile_rows=tile_rows)
    try:
        sp = e2e.c2_params(seed)
        synth(lib_ctx, rf, sp)
        fill_pictures(rf, seed)
        cf = rf.array("cf", np.uint8)
        out = {}
        for t in threads:
            cf[:] = rf.cf_copy
            t0 = time.perf_counter()
            rf.recon(t)
            out[int(t)] = round(w * h / (time.perf_counter() - t0) / 1e6, 1)
        return out
    finally:
        rf.destroy()


def full_route_rate(ctx, w, h, bpc, tile_cols=16, tile_rows=8, threads=64, frames=5, seed=0xF0E, progress=None):
    """The whole frame from dav1d's hand-off to final pixels, timed: pass-1 arrays (Av1Block / cbi / cf) and pass 1's filter inputs
    (Av1Filter masks, level cache, cdef_idx, restoration units — built here by the reference's own dav1d_create_lf_mask_* on a real
    Dav1dFrameContext) -> lister threads -> filter lister threads -> dav1d_hip_frame_end (reconstruction, deblocking, CDEF,
    restoration).  Per frame the dense coefficient arena and the level cache cross the host link while the listing runs.  The
    final picture is compared with the reference's dav1d_decode_tile_sbrow + dav1d_filter_sbrow of the same frame.  Returns the
    measurement dict, or None without the reference build."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    import e2e
    if ref_lib() is None:
        return None
    filters = dict(lf=(20, 28, 16, 24, 0, False), cdef=(5, 2, [17, 33, 0, 63], [5, 0, 20, 48]), lr=([1, 1, 1], [6, 6]))
    rf = RefFrame(w, h, 1, bpc, is_inter=True, sb128=True, tile_cols=tile_cols, tile_rows=tile_rows, filters=filters)
    try:
        sp = e2e.c2_params(seed)
        d = synth(ctx, rf, sp)
        fill_pictures(rf, seed + 1)
        rf.build_filter_inputs(seed)
        t0 = time.perf_counter()
        rf.recon(min(threads, 64))
        t_ref_recon = time.perf_counter() - t0
        t0 = time.perf_counter()
        rf.filter()
        t_ref_filter = time.perf_counter() - t0
        n_pl = 3
        cur = ctx.picture(w, h, 1, bpc)
        refs = []
        for i in range(7):
            r = ctx.picture(rf.p.ref_w[i], rf.p.ref_h[i], 1, bpc)
            for pl in range(n_pl):
                rows, cols = r.padded_shape(pl)
                r.upload(pl, np.ascontiguousarray(rf.plane(1 + i, pl)[:rows, :cols]))
            refs.append(r)
        coef = ctx.buffer(len(rf.cf_copy))
        lvl_host = rf.array("lf_level", np.uint8)
        lvl = ctx.buffer(len(lvl_host))
        lut = rf.array("lim_lut", np.uint8)
        fd = rf.filter_desc()
        nb = C.c_size_t()
        blob = ctx.lib.dav1d_hip_lister_const_masks(C.byref(nb))
        prep = mask = None
        res = {"list_ms": [], "filter_list_ms": [], "h2d_ms": [], "frame_end_ms": [], "total_ms": []}
        planes = None
        with ThreadPoolExecutor(1) as ex2:
            for it in range(frames):
                t_a = time.perf_counter()
                frame = ctx.frame(cur, refs)
                lh = C.c_void_p()
                assert ctx.lib.dav1d_hip_lister_create(C.byref(lh), C.byref(d), frame.h) == 0

                def h2d():
                    t = time.perf_counter()
                    coef.upload(rf.cf_copy)
                    lvl.upload(lvl_host)
                    return (time.perf_counter() - t) * 1e3
                up = ex2.submit(h2d)
                assert ctx.lib.dav1d_hip_lister_run(lh, threads) == 0
                t_b = time.perf_counter()
                assert ctx.lib.dav1d_hip_lister_filter_run(lh, C.byref(fd), threads) == 0
                t_c = time.perf_counter()
                h2d_ms = up.result()
                if prep is None:
                    prep = ctx.buffer(ctx.lib.dav1d_hip_lister_prep_elems(lh) * 2 + 4096)
                    mask = ctx.buffer(ctx.lib.dav1d_hip_lister_mask_bytes(lh) + 4096)
                    mask.upload(np.ctypeslib.as_array((C.c_uint8 * nb.value).from_address(blob)))
                frame.set_filters(lvl, rf.b4_stride, lut[0:64], lut[64:128], rf.p.cdef_damping + bpc - 8)
                if progress is not None:
                    frame.set_progress_callback(lambda rows, pic: progress.append((it, rows)))      # a listener: the last stage runs in row bands
                t_d = time.perf_counter()
                filtered = frame.end(coef, prep, mask)
                t_e = time.perf_counter()
                if it == frames - 1:
                    fpic = api.DevicePicture.view(ctx, filtered, w, h, 1, bpc)
                    planes = [fpic.download(pl) for pl in range(n_pl)]
                ctx.lib.dav1d_hip_lister_destroy(lh)
                frame.destroy()
                if it:
                    res["list_ms"].append((t_b - t_a) * 1e3)
                    res["filter_list_ms"].append((t_c - t_b) * 1e3)
                    res["h2d_ms"].append(h2d_ms)
                    res["frame_end_ms"].append((t_e - t_d) * 1e3)
                    res["total_ms"].append((t_e - t_a) * 1e3)
        bad = compare(rf, planes)
        if bad:
            raise AssertionError("full route: filtered planes differ from the reference's: %s" % bad)
        out = {k: round(float(np.median(v)), 3) for k, v in res.items() if v}
        out.update(frames=frames - 1, host_threads=threads, tiles=tile_cols * tile_rows,
                   value=round(w * h / (out["total_ms"] * 1e-3) / 1e6, 1), unit="Mpixels/s",
                   parity="bit-exact vs the reference's dav1d_decode_tile_sbrow + dav1d_filter_sbrow (%.2f s + %.2f s on this host)" % (t_ref_recon, t_ref_filter),
                   workload="%dx%d 4:2:0 %d-bit inter frame, hand-off arrays + pass 1's filter inputs -> %d library threads listing blocks, then "
                            "filter tasks -> reconstruction, deblocking (levels 20/28/16/24), CDEF (4 strength pairs), switchable restoration "
                            "(64-pixel units); coefficients (dense) and level cache cross the host link every frame" % (w, h, bpc, threads))
        for o in refs + [cur, coef, lvl] + ([prep, mask] if prep is not None else []):
            o.free()
        return out
    finally:
        rf.destroy()


def row_progress_cost(ctx, w, h, bpc, tile_cols=16, tile_rows=8, threads=64, frames=7):
    """What row-granular progress costs (dav1d_hip_frame_set_progress_callback on the stage-by-stage schedule: the last stage in bands of
    256 rows, an event behind each, rows published while the later bands run): dav1d_hip_frame_end of the same frame with and without
    a listener, pictures checked against the reference both times."""
    calls = []
    a = full_route_rate(ctx, w, h, bpc, tile_cols, tile_rows, threads, frames)
    b = full_route_rate(ctx, w, h, bpc, tile_cols, tile_rows, threads, frames, progress=calls)
    if not a or not b:
        return None
    per_frame = {}
    for it, rows in calls:
        per_frame.setdefault(it, []).append(rows)
    last = per_frame[max(per_frame)]
    return {"frame_end_ms": a["frame_end_ms"], "frame_end_ms_with_listener": b["frame_end_ms"],
            "cost_pct": round((b["frame_end_ms"] / a["frame_end_ms"] - 1) * 100, 1), "publications_per_frame": len(last),
            "rows_published": last, "parity": b["parity"]}


def full_route_sustained(ctx, w, h, bpc, tile_cols=16, tile_rows=8, threads=None, frames=10, depth=2, warm=2, seed=0xF0E):
    """full_route_rate with frames in flight (e2e.run_pipelined): frame n + 1 is listed — blocks by the packing lister,
    then the filter tasks — while frame n runs on the device; per frame only the coefficients that exist, the prepared lists and the
    level cache cross the host link.  The last frame's final picture is compared with the reference's dav1d_decode_tile_sbrow +
    dav1d_filter_sbrow.  Returns the measurement dict, or None without the reference build."""
    import e2e
    if ref_lib() is None:
        return None
    filters = dict(lf=(20, 28, 16, 24, 0, False), cdef=(5, 2, [17, 33, 0, 63], [5, 0, 20, 48]), lr=([1, 1, 1], [6, 6]))
    rf = RefFrame(w, h, 1, bpc, is_inter=True, sb128=True, tile_cols=tile_cols, tile_rows=tile_rows, filters=filters)
    try:
        sp = e2e.c2_params(seed)
        d = synth(ctx, rf, sp)
        fill_pictures(rf, seed + 1)
        rf.build_filter_inputs(seed)
        threads = e2e.host_threads(tile_cols * tile_rows + 3 * rf.sbh, threads)
        rf.recon(min(threads, 64))
        rf.filter()
        refs = []
        for i in range(7):
            r = ctx.picture(rf.p.ref_w[i], rf.p.ref_h[i], 1, bpc)
            for pl in range(3):
                rows, cols = r.padded_shape(pl)
                r.upload(pl, np.ascontiguousarray(rf.plane(1 + i, pl)[:rows, :cols]))
            refs.append(r)
        lut = rf.array("lim_lut", np.uint8)
        flt = dict(fd=rf.filter_desc(), lvl_host=rf.array("lf_level", np.uint8), b4_stride=rf.b4_stride, lut_e=lut[0:64], lut_i=lut[64:128],
                   damping=rf.p.cdef_damping + bpc - 8)
        cfs = [rf.cf_copy.copy() for _ in range(frames)]
        out, planes = e2e.run_pipelined(ctx, d, cfs, w, h, 1, bpc, refs, threads, depth, warm, filters=flt)
        bad = compare(rf, planes)
        if bad:
            raise AssertionError("full route, frames in flight: filtered planes differ from the reference's: %s" % bad)
        out.update(tiles=tile_cols * tile_rows, host_arena_left_zero=not any(bool(c.any()) for c in cfs),
                   parity="bit-exact vs the reference's dav1d_decode_tile_sbrow + dav1d_filter_sbrow (last of the frames)",
                   workload="%dx%d 4:2:0 %d-bit inter frame, %d x %d tiles, %d frames in flight: hand-off arrays + pass 1's filter inputs -> %d library "
                            "threads (packing lister, then filter tasks) while the frame before runs reconstruction, deblocking, CDEF, switchable "
                            "restoration; eob + 1 values per block, the prepared lists and the level cache cross the host link" %
                            (w, h, bpc, tile_cols, tile_rows, depth, threads))
        for o in refs:
            o.free()
        return out
    finally:
        rf.destroy()


def c0_line(ctx, w=1920, h=1080, bpc=8, seed=0xC0, strict=True):
    """BASELINE configs[0] (1080p 8-bit, 64-pixel superblocks, one tile, ONE CPU thread; the Chimera stream itself does not
    exist here, a synthetic inter frame of every tool stands in): the reference's own pass 2 + in-loop filters on one thread — the
    plumbing baseline — and the SAME reference code with its DSP table replaced by the reference-signature table of the HIP library
    (dav1d_hip_dsp_init_8bpc, INTEGRATION.md 1: every DSP call staged to the device, run by the batched kernel on one task, copied
    back).  Both pictures must be identical."""
    import time
    if ref_lib() is None:
        return None
    filters = dict(lf=(20, 28, 16, 24, 0, False), cdef=(5, 2, [17, 33, 0, 63], [5, 0, 20, 48]), lr=([1, 1, 1], [6, 6]))
    out = {}
    pics = []
    n_slots = 421
    for leg in ("reference C, 1 thread", "reference drivers + HIP DSP table"):
        rf = RefFrame(w, h, 1, bpc, is_inter=True, sb128=False, tile_cols=1, tile_rows=1, filters=filters)
        try:
            sp = default_synth(seed, n_refs=3, far_mv_pct=2)
            synth(ctx, rf, sp)
            fill_pictures(rf, seed + 1)
            rf.build_filter_inputs(seed)
            if leg != "reference C, 1 thread":
                tab = (C.c_void_p * n_slots)()
                rc = ctx.lib.dav1d_hip_dsp_init_8bpc(tab) if bpc == 8 else ctx.lib.dav1d_hip_dsp_init_16bpc(tab, bpc)
                assert rc == 0, rc
                rf.lib.dav1d_ref_frame_use_dsp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
                assert rf.lib.dav1d_ref_frame_use_dsp(rf.h, tab, C.sizeof(tab)) == 0
            t0 = time.perf_counter()
            rf.recon(1)
            t1 = time.perf_counter()
            rf.filter()
            t2 = time.perf_counter()
            pics.append([rf.plane(0, pl).copy() for pl in range(3)])
            out[leg] = {"value": round(w * h / (t2 - t0) / 1e6, 2), "unit": "Mpixels/s", "recon_s": round(t1 - t0, 3), "filters_s": round(t2 - t1, 3)}
        finally:
            rf.destroy()
    parity = "bit-exact: the picture through the HIP DSP table equals the reference C picture"
    for pl in range(3):
        if not np.array_equal(pics[0][pl], pics[1][pl]):
            bad = np.argwhere(pics[0][pl] != pics[1][pl])
            parity = ("MISMATCH: plane %d through the HIP DSP table differs from the reference C: %d pixels, first (y, x) %s, rows %d..%d, columns %d..%d"
                      % (pl, len(bad), bad[0].tolist(), bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max()))
            if strict:
                raise AssertionError("C0: " + parity)
            break
    return {"workload": "%dx%d 4:2:0 %d-bit inter frame, 64-pixel superblocks, one tile, every prediction tool, deblock + CDEF + restoration; "
                        "dav1d_decode_tile_sbrow + dav1d_filter_sbrow on ONE host thread" % (w, h, bpc),
            "cpu_c_1_thread": out["reference C, 1 thread"],
            "dsp_table_drop_in": dict(out["reference drivers + HIP DSP table"], what="the same reference code, every DSP call through dav1d_hip_dsp_init's table "
                                      "(one staged task per call: the bring-up path, not the fast one)"),
            "parity": parity}
