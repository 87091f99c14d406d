"""Thin object layer over the C ABI for Python callers (tests, bench.py, smoke()).

It only moves numpy arrays / device pointers in and out of the C entry points of
include/dav1d_hip.h; all reconstruction work happens in the HIP library.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FilmGrainData  # noqa: F401
from ._lib import ITX_TASK, MC_TASK, COMP_TASK, CDEF_TASK, LF_TASK, IPRED_TASK, LR_TASK, WARP_TASK, MC_SCALED_TASK, Picture, HostPicture  # noqa: F401  (re-exported)

LAYOUT_I400, LAYOUT_I420, LAYOUT_I422, LAYOUT_I444 = 0, 1, 2, 3


class HipError(RuntimeError):
    pass


def _chk(rc, what):
    if rc:
        why = ""
        if rc in (-5, -12, -38):          # -EIO / -ENOMEM / -ENOSYS: a HIP call was behind it
            try:
                why = " (HIP: %s)" % _lib.load().dav1d_hip_last_hip_error(None).decode()
            except Exception:           # noqa: BLE001
                pass
        raise HipError("%s failed: errno %d%s" % (what, -rc, why))


class DeviceBuffer:
    """A device allocation made through dav1d_hip_malloc."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        _chk(ctx.lib.dav1d_hip_malloc(ctx.h, C.byref(p), self.nbytes), "malloc")
        self.ptr = p.value

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        _chk(self.ctx.lib.dav1d_hip_upload(self.ctx.h, self.ptr + offset, a.ctypes.data, a.nbytes), "upload")

    def download(self, dtype, count=None, offset=0):
        dt = np.dtype(dtype)
        n = (self.nbytes - offset) // dt.itemsize if count is None else count
        out = np.empty(n, dt)
        _chk(self.ctx.lib.dav1d_hip_download(self.ctx.h, out.ctypes.data, self.ptr + offset, out.nbytes), "download")
        return out

    def zero(self):
        _chk(self.ctx.lib.dav1d_hip_memset(self.ctx.h, self.ptr, 0, self.nbytes), "memset")

    def free(self):
        if self.ptr:
            self.ctx.lib.dav1d_hip_free(self.ctx.h, self.ptr)
            self.ptr = None


TWIN_ONLY = 2          # Dav1dHipPicture.twin_ok: the picture lives in its tiled twin, the raster planes are stale


class DevicePicture:
    def __init__(self, ctx, w, h, layout, bpc):
        self.ctx = ctx
        self.pic = Picture()
        _chk(ctx.lib.dav1d_hip_picture_alloc(ctx.h, C.byref(self.pic), w, h, layout, bpc), "picture_alloc")
        self.w, self.h, self.layout, self.bpc = w, h, layout, bpc
        self.dtype = np.uint8 if bpc == 8 else np.uint16

    @property
    def n_planes(self):
        return 1 if self.layout == LAYOUT_I400 else 3

    def padded_shape(self, plane):
        ss_ver = 1 if plane and self.layout == LAYOUT_I420 else 0
        ss_hor = 1 if plane and self.layout != LAYOUT_I444 else 0
        return (((self.h + 127) & ~127) >> ss_ver, ((self.w + 127) & ~127) >> ss_hor)

    def stride_px(self, plane):
        return self.pic.p[plane].stride // np.dtype(self.dtype).itemsize

    def upload(self, plane, arr):
        """arr: 2-D array of the PADDED plane shape (rows x cols)."""
        a = np.asarray(arr, dtype=self.dtype)
        if a.strides[1] != a.itemsize:
            a = np.ascontiguousarray(a)
        assert a.shape == self.padded_shape(plane), (a.shape, self.padded_shape(plane))
        _chk(self.ctx.lib.dav1d_hip_plane_upload(self.ctx.h, C.byref(self.pic), plane, a.ctypes.data,
                                                 a.strides[0], 1), "plane_upload")
        # the raster planes changed behind the tiled twin's back (Dav1dHipPicture.twin_ok is the caller's to keep)
        self.pic.twin_ok = 0
        if getattr(self.ctx, "auto_retile", False) and plane == self.n_planes - 1:
            self.retile()

    def untile(self):
        """dav1d_hip_picture_untile: a picture that lives in its twin gets its raster planes back (no-op otherwise)."""
        _chk(self.ctx.lib.dav1d_hip_picture_untile(self.ctx.h, C.byref(self.pic)), "picture_untile")

    def retile(self, overlapped=False):
        """(Re)build the tiled twin from the raster planes: motion compensation then reads this picture through it.  overlapped: on a
        side stream, next to whatever is enqueued afterwards (dav1d_hip_picture_retile_overlapped)."""
        fn = self.ctx.lib.dav1d_hip_picture_retile_overlapped if overlapped else self.ctx.lib.dav1d_hip_picture_retile
        _chk(fn(self.ctx.h, C.byref(self.pic)), "picture_retile")

    def download(self, plane):
        """Padded plane as a (rows x cols) view of a host array with the DEVICE row stride, so that
        the same pixel offsets address host and device copies."""
        rows, cols = self.padded_shape(plane)
        out = np.zeros((rows, self.stride_px(plane)), self.dtype)[:, :cols]
        _chk(self.ctx.lib.dav1d_hip_plane_download(self.ctx.h, C.byref(self.pic), plane, out.ctypes.data,
                                                   out.strides[0], 1), "plane_download")
        return out

    @classmethod
    def view(cls, ctx, pic, w, h, layout, bpc):
        """A non-owning wrapper around a Picture descriptor (e.g. the frame-owned output of FrameInFlight.end())."""
        self = cls.__new__(cls)
        self.ctx, self.pic = ctx, pic
        self.w, self.h, self.layout, self.bpc = w, h, layout, bpc
        self.dtype = np.uint8 if bpc == 8 else np.uint16
        self.borrowed = True
        return self

    def free(self):
        if not getattr(self, "borrowed", False):
            self.ctx.lib.dav1d_hip_picture_free(self.ctx.h, C.byref(self.pic))


class _List:
    def __init__(self, ctx, kind, tasks, dtype):
        self.ctx, self.kind = ctx, kind
        t = np.ascontiguousarray(tasks, dtype=dtype)
        self.n = len(t)
        self.h = C.c_void_p()
        _chk(getattr(ctx.lib, "dav1d_hip_%s_list_create" % kind)(ctx.h, C.byref(self.h), t.ctypes.data, len(t)),
             kind + "_list_create")

    def destroy(self):
        if self.h:
            getattr(self.ctx.lib, "dav1d_hip_%s_list_destroy" % self.kind)(self.ctx.h, self.h)
            self.h = None


class _InterList:
    def __init__(self, ctx, mc_tasks, comp_tasks):
        self.ctx = ctx
        m = np.ascontiguousarray(mc_tasks, dtype=MC_TASK)
        k = np.ascontiguousarray(comp_tasks, dtype=COMP_TASK)
        self.h = C.c_void_p()
        _chk(ctx.lib.dav1d_hip_inter_list_create(ctx.h, C.byref(self.h), m.ctypes.data, len(m), k.ctypes.data, len(k)),
             "inter_list_create")
        self.n_fused = int(ctx.lib.dav1d_hip_inter_list_fused(self.h))

    def destroy(self):
        if self.h:
            self.ctx.lib.dav1d_hip_inter_list_destroy(self.ctx.h, self.h)
            self.h = None


class _ReconList:
    """dav1d_hip_recon_list_*: predictions + residuals of a frame, pipelined per tile shape / transform size."""

    def __init__(self, ctx, geometry, mc_tasks, comp_tasks, itx_tasks):
        self.ctx = ctx
        m = np.ascontiguousarray(mc_tasks, dtype=MC_TASK)
        k = np.ascontiguousarray(comp_tasks, dtype=COMP_TASK)
        t = np.ascontiguousarray(itx_tasks, dtype=ITX_TASK)
        self.h = C.c_void_p()
        _chk(ctx.lib.dav1d_hip_recon_list_create(ctx.h, C.byref(self.h), C.byref(geometry.pic), m.ctypes.data, len(m),
                                                 k.ctypes.data, len(k), t.ctypes.data, len(t)), "recon_list_create")

    def run(self, dst, refs, prep, coef, mask=None):
        if getattr(self.ctx, "tiled_native", False):       # (tests: every recon list of the context leaves its picture in the twin only)
            return self.run_tiled(dst, refs, prep, coef, mask)
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        p = None if prep is None else (prep.ptr if hasattr(prep, "ptr") else prep)
        m = None if mask is None else (mask.ptr if hasattr(mask, "ptr") else mask)
        _chk(self.ctx.lib.dav1d_hip_recon_list_run(self.ctx.h, self.h, C.byref(dst.pic), arr, len(refs), p, m,
                                                   coef.ptr if hasattr(coef, "ptr") else coef), "recon_list_run")

    def run_tiled(self, dst, refs, prep, coef, mask=None):
        """dav1d_hip_recon_list_run_tiled: the picture lives in its tiled twin only afterwards (dst.pic.twin_ok == TWIN_ONLY when every
        launch could write tiles; 1 after the raster + retile fallback).  download() / the fetch calls un-tile on the way out."""
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        p = None if prep is None else (prep.ptr if hasattr(prep, "ptr") else prep)
        m = None if mask is None else (mask.ptr if hasattr(mask, "ptr") else mask)
        _chk(self.ctx.lib.dav1d_hip_recon_list_run_tiled(self.ctx.h, self.h, C.byref(dst.pic), arr, len(refs), p, m,
                                                         coef.ptr if hasattr(coef, "ptr") else coef), "recon_list_run_tiled")

    def run_twin(self, dst, refs, prep, coef, mask=None):
        """dav1d_hip_recon_list_run_twin: the frame's pixels also end up in dst's tiled twin (written by the launches themselves when
        they can, by a retile pass otherwise); dst.pic.twin_ok is set."""
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        p = None if prep is None else (prep.ptr if hasattr(prep, "ptr") else prep)
        m = None if mask is None else (mask.ptr if hasattr(mask, "ptr") else mask)
        _chk(self.ctx.lib.dav1d_hip_recon_list_run_twin(self.ctx.h, self.h, C.byref(dst.pic), arr, len(refs), p, m,
                                                        coef.ptr if hasattr(coef, "ptr") else coef), "recon_list_run_twin")

    def destroy(self):
        if self.h:
            self.ctx.lib.dav1d_hip_recon_list_destroy(self.ctx.h, self.h)
            self.h = None


class Context:
    """dav1d_hip_open() wrapper.  `stream` is a raw hipStream_t (int) or None."""

    def __init__(self, device=0, stream=None, lib_path=None):
        self.lib = _lib.load(lib_path)
        self.device, self.lib_path = device, lib_path
        h = C.c_void_p()
        rc = self.lib.dav1d_hip_open(C.byref(h), device, stream)
        if rc:
            raise HipError("dav1d_hip_open(device=%d) failed: errno %d (no usable MI355X device; there is no "
                           "CPU fallback)" % (device, -rc))
        self.h = h

    def close(self):
        if self.h:
            self.lib.dav1d_hip_close(self.h)
            self.h = None

    def get_option(self, name):
        """dav1d_hip_get_option: a counter or knob of this context (see include/dav1d_hip.h)"""
        v = C.c_long()
        _chk(self.lib.dav1d_hip_get_option(self.h, name.encode(), C.byref(v)), "get_option(%s)" % name)
        return v.value

    def set_option(self, name, value):
        """dav1d_hip_set_option: a tuning knob of this context (see include/dav1d_hip.h)"""
        _chk(self.lib.dav1d_hip_set_option(self.h, name.encode(), int(value)), "set_option(%s)" % name)

    def sync(self):
        _chk(self.lib.dav1d_hip_sync(self.h), "sync")

    def buffer(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def buffer_from(self, arr):
        a = np.ascontiguousarray(arr)
        b = DeviceBuffer(self, max(a.nbytes, 16))
        if a.nbytes:
            b.upload(a)
        return b

    def picture(self, w, h, layout, bpc):
        return DevicePicture(self, w, h, layout, bpc)

    def graph_begin(self):
        """Start recording the list runs issued on this context (dav1d_hip_graph_begin)."""
        _chk(self.lib.dav1d_hip_graph_begin(self.h), "graph_begin")

    def graph_end(self):
        g = C.c_void_p()
        _chk(self.lib.dav1d_hip_graph_end(self.h, C.byref(g)), "graph_end")
        return g

    def graph_launch(self, g):
        _chk(self.lib.dav1d_hip_graph_launch(self.h, g), "graph_launch")

    def graph_destroy(self, g):
        self.lib.dav1d_hip_graph_destroy(self.h, g)

    def last_kernel_ms(self):
        return float(self.lib.dav1d_hip_last_kernel_ms(self.h))

    # ---- batched entry points (host task arrays, device arenas)
    def itx_add_batch(self, dst, tasks, coef):
        t = np.ascontiguousarray(tasks, dtype=ITX_TASK)
        _chk(self.lib.dav1d_hip_itx_add_batch(self.h, C.byref(dst.pic), t.ctypes.data, len(t), coef.ptr), "itx_add_batch")

    def mc_batch(self, dst, refs, tasks, prep=None):
        t = np.ascontiguousarray(tasks, dtype=MC_TASK)
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        _chk(self.lib.dav1d_hip_mc_batch(self.h, C.byref(dst.pic), arr, len(refs), t.ctypes.data, len(t),
                                         prep.ptr if prep else None), "mc_batch")

    def comp_batch(self, dst, tasks, prep, mask=None):
        t = np.ascontiguousarray(tasks, dtype=COMP_TASK)
        _chk(self.lib.dav1d_hip_comp_batch(self.h, C.byref(dst.pic), t.ctypes.data, len(t), prep.ptr,
                                           mask.ptr if mask else None), "comp_batch")

    def warp_batch(self, dst, refs, tasks, prep=None):
        t = np.ascontiguousarray(tasks, dtype=WARP_TASK)
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        _chk(self.lib.dav1d_hip_warp_batch(self.h, C.byref(dst.pic), arr, len(refs), t.ctypes.data, len(t),
                                           prep.ptr if prep else None), "warp_batch")

    def mc_scaled_batch(self, dst, refs, tasks, prep=None):
        t = np.ascontiguousarray(tasks, dtype=MC_SCALED_TASK)
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        _chk(self.lib.dav1d_hip_mc_scaled_batch(self.h, C.byref(dst.pic), arr, len(refs), t.ctypes.data, len(t),
                                                prep.ptr if prep else None), "mc_scaled_batch")

    def resize(self, dst, src, plane, dst_w, y0, h, src_w, dx, mx0):
        _chk(self.lib.dav1d_hip_resize(self.h, C.byref(dst.pic), C.byref(src.pic), plane, dst_w, y0, h, src_w, dx, mx0), "resize")

    def emu_edge(self, bpc, bw, bh, iw, ih, x, y, dst, dst_stride, ref, ref_stride):
        """dst / ref: DeviceBuffer (or raw device address); strides in bytes."""
        d = dst.ptr if hasattr(dst, "ptr") else dst
        r = ref.ptr if hasattr(ref, "ptr") else ref
        _chk(self.lib.dav1d_hip_emu_edge(self.h, bpc, bw, bh, iw, ih, x, y, d, dst_stride, r, ref_stride), "emu_edge")

    def cdef_batch(self, dst, src, tasks, damping, dirvar=None):
        t = np.ascontiguousarray(tasks, dtype=CDEF_TASK)
        _chk(self.lib.dav1d_hip_cdef_batch(self.h, C.byref(dst.pic), C.byref(src.pic), t.ctypes.data, len(t), damping,
                                           dirvar.ptr if dirvar else None), "cdef_batch")

    def lf_batch(self, dst, tasks, lvl, b4_stride, lut_e, lut_i):
        t = np.ascontiguousarray(tasks, dtype=LF_TASK)
        e = np.ascontiguousarray(lut_e, dtype=np.uint8)
        i = np.ascontiguousarray(lut_i, dtype=np.uint8)
        assert len(e) == 64 and len(i) == 64
        _chk(self.lib.dav1d_hip_lf_batch(self.h, C.byref(dst.pic), t.ctypes.data, len(t), lvl.ptr, b4_stride,
                                         e.ctypes.data, i.ctypes.data), "lf_batch")

    def ipred_batch(self, dst, tasks, pal_idx=None):
        t = np.ascontiguousarray(tasks, dtype=IPRED_TASK)
        _chk(self.lib.dav1d_hip_ipred_batch(self.h, C.byref(dst.pic), t.ctypes.data, len(t),
                                            pal_idx.ptr if pal_idx else None), "ipred_batch")

    def lr_batch(self, dst, src, lpf, tasks):
        t = np.ascontiguousarray(tasks, dtype=LR_TASK)
        _chk(self.lib.dav1d_hip_lr_batch(self.h, C.byref(dst.pic), C.byref(src.pic), C.byref(lpf.pic), t.ctypes.data, len(t)),
             "lr_batch")

    def fg_apply(self, dst, src, data, is_id=0):
        _chk(self.lib.dav1d_hip_fg_apply(self.h, C.byref(dst.pic), C.byref(src.pic), C.byref(data), is_id), "fg_apply")

    def fg_prepare(self, data, bpc, layout):
        """dav1d_hip_fg_prepare: templates + scaling tables on a side stream; returns the handle for fg_apply_prepared."""
        g = C.c_void_p()
        _chk(self.lib.dav1d_hip_fg_prepare(self.h, C.byref(g), C.addressof(data), bpc, layout), "fg_prepare")
        return g

    def fg_apply_prepared(self, dst, src, g, is_id=0):
        _chk(self.lib.dav1d_hip_fg_apply_prepared(self.h, C.byref(dst.pic), C.byref(src.pic), g, is_id), "fg_apply_prepared")

    def fg_grain_destroy(self, g):
        self.lib.dav1d_hip_fg_grain_destroy(self.h, g)

    def fg_generate_grain(self, data, bpc, layout):
        out = np.zeros((3, 74, 82), np.int16)
        _chk(self.lib.dav1d_hip_fg_generate_grain(self.h, C.byref(data), bpc, layout, out.ctypes.data), "fg_generate_grain")
        return out

    def frame(self, cur, refs):
        return FrameInFlight(self, cur, refs)

    def intra_list(self, batches):
        return _IntraList(self, batches)

    def ipred_list(self, batches):
        return _IpredList(self, batches)

    def intra_flow(self, batches):
        return _IntraFlow(self, batches)

    def intra_sb(self, batches, geometry, sb128=False, col_start_sb=None, row_start_sb=None):
        return _IntraSb(self, batches, geometry, sb128, col_start_sb, row_start_sb)

    # ---- device-resident lists
    def itx_list(self, tasks):
        return _List(self, "itx", tasks, ITX_TASK)

    def mc_list(self, tasks):
        return _List(self, "mc", tasks, MC_TASK)

    def comp_list(self, tasks):
        return _List(self, "comp", tasks, COMP_TASK)

    def recon_list(self, geometry, mc_tasks, comp_tasks, itx_tasks):
        return _ReconList(self, geometry, mc_tasks, comp_tasks, itx_tasks)

    def inter_list(self, mc_tasks, comp_tasks):
        return _InterList(self, mc_tasks, comp_tasks)

    def run_inter_list(self, lst, dst, refs, prep=None, mask=None):
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        p = None if prep is None else (prep.ptr if hasattr(prep, "ptr") else prep)
        m = None if mask is None else (mask.ptr if hasattr(mask, "ptr") else mask)
        _chk(self.lib.dav1d_hip_inter_list_run(self.h, lst.h, C.byref(dst.pic), arr, len(refs), p, m), "inter_list_run")

    def run_itx_list(self, lst, dst, coef):
        _chk(self.lib.dav1d_hip_itx_list_run(self.h, lst.h, C.byref(dst.pic), coef.ptr if hasattr(coef, "ptr") else coef),
             "itx_list_run")

    def run_mc_list(self, lst, dst, refs, prep=None):
        arr = (Picture * len(refs))(*[r.pic for r in refs])
        p = None if prep is None else (prep.ptr if hasattr(prep, "ptr") else prep)
        _chk(self.lib.dav1d_hip_mc_list_run(self.h, lst.h, C.byref(dst.pic), arr, len(refs), p), "mc_list_run")

    def run_comp_list(self, lst, dst, prep, mask=None):
        p = prep.ptr if hasattr(prep, "ptr") else prep
        m = None if mask is None else (mask.ptr if hasattr(mask, "ptr") else mask)
        _chk(self.lib.dav1d_hip_comp_list_run(self.h, lst.h, C.byref(dst.pic), p, m), "comp_list_run")


class _IpredList:
    """dav1d_hip_ipred_list_*: the intra batches of a wavefront, device resident."""

    def __init__(self, ctx, batches):
        self.ctx = ctx
        self.n_batches = len(batches)
        sizes = (C.c_size_t * max(len(batches), 1))(*[len(b) for b in batches])
        allt = np.ascontiguousarray(np.concatenate(batches) if len(batches) else np.zeros(0, IPRED_TASK), dtype=IPRED_TASK)
        self.h = C.c_void_p()
        _chk(ctx.lib.dav1d_hip_ipred_list_create(ctx.h, C.byref(self.h), allt.ctypes.data, sizes, len(batches)), "ipred_list_create")

    def run_batch(self, k, dst, aux=None):
        _chk(self.ctx.lib.dav1d_hip_ipred_list_run_batch(self.ctx.h, self.h, k, C.byref(dst.pic), aux.ptr if aux else None), "ipred_list_run_batch")

    def destroy(self):
        if self.h:
            self.ctx.lib.dav1d_hip_ipred_list_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class _IntraList:
    """dav1d_hip_intra_list_*: predictions + residuals of every wavefront step (small blocks paired in one wave)."""

    def __init__(self, ctx, batches):
        """batches: [(ipred tasks, itx tasks)] per wavefront step."""
        self.ctx = ctx
        self.n_batches = len(batches)
        ps = (C.c_size_t * max(len(batches), 1))(*[len(b[0]) for b in batches])
        ts = (C.c_size_t * max(len(batches), 1))(*[len(b[1]) for b in batches])
        allp = np.ascontiguousarray(np.concatenate([b[0] for b in batches]) if batches else np.zeros(0, IPRED_TASK), dtype=IPRED_TASK)
        allt = np.ascontiguousarray(np.concatenate([b[1] for b in batches]) if batches else np.zeros(0, ITX_TASK), dtype=ITX_TASK)
        self.h = C.c_void_p()
        _chk(ctx.lib.dav1d_hip_intra_list_create(ctx.h, C.byref(self.h), allp.ctypes.data, ps, allt.ctypes.data, ts, len(batches)),
             "intra_list_create")

    def run_batch(self, k, dst, coef, aux=None):
        _chk(self.ctx.lib.dav1d_hip_intra_list_run_batch(self.ctx.h, self.h, k, C.byref(dst.pic), coef.ptr if hasattr(coef, "ptr") else coef,
                                                         aux.ptr if aux else None), "intra_list_run_batch")

    def run_all(self, dst, coef, aux=None):
        _chk(self.ctx.lib.dav1d_hip_intra_list_run_all(self.ctx.h, self.h, C.byref(dst.pic), coef.ptr if hasattr(coef, "ptr") else coef,
                                                       aux.ptr if aux else None), "intra_list_run_all")

    def destroy(self):
        if self.h:
            self.ctx.lib.dav1d_hip_intra_list_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class _IntraFlow:
    """dav1d_hip_intra_flow_*: the whole wavefront as one launch."""

    def __init__(self, ctx, batches):
        self.ctx = ctx
        ps = (C.c_size_t * max(len(batches), 1))(*[len(b[0]) for b in batches])
        ts = (C.c_size_t * max(len(batches), 1))(*[len(b[1]) for b in batches])
        allp = np.ascontiguousarray(np.concatenate([b[0] for b in batches]) if batches else np.zeros(0, IPRED_TASK), dtype=IPRED_TASK)
        allt = np.ascontiguousarray(np.concatenate([b[1] for b in batches]) if batches else np.zeros(0, ITX_TASK), dtype=ITX_TASK)
        self.h = C.c_void_p()
        _chk(ctx.lib.dav1d_hip_intra_flow_create(ctx.h, C.byref(self.h), allp.ctypes.data, ps, allt.ctypes.data, ts, len(batches)),
             "intra_flow_create")
        self.n_units = int(ctx.lib.dav1d_hip_intra_flow_units(self.h))

    def run(self, dst, coef, aux=None):
        _chk(self.ctx.lib.dav1d_hip_intra_flow_run(self.ctx.h, self.h, C.byref(dst.pic), coef.ptr if hasattr(coef, "ptr") else coef,
                                                   aux.ptr if aux else None), "intra_flow_run")

    def status(self):
        out = (C.c_uint32 * 3)()
        _chk(self.ctx.lib.dav1d_hip_intra_flow_status(self.ctx.h, self.h, out), "intra_flow_status")
        return tuple(int(v) for v in out)

    def destroy(self):
        if self.h:
            self.ctx.lib.dav1d_hip_intra_flow_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class _IntraSb:
    """dav1d_hip_intra_sb_*: the wavefront superblock by superblock (a workgroup per superblock, a launch per level).
    col_start_sb / row_start_sb: the tile starts in superblocks, last entry = the end (None: one tile)."""

    def __init__(self, ctx, batches, geometry, sb128=False, col_start_sb=None, row_start_sb=None):
        self.ctx = ctx
        ps = (C.c_size_t * max(len(batches), 1))(*[len(b[0]) for b in batches])
        ts = (C.c_size_t * max(len(batches), 1))(*[len(b[1]) for b in batches])
        allp = np.ascontiguousarray(np.concatenate([b[0] for b in batches]) if batches else np.zeros(0, IPRED_TASK), dtype=IPRED_TASK)
        allt = np.ascontiguousarray(np.concatenate([b[1] for b in batches]) if batches else np.zeros(0, ITX_TASK), dtype=ITX_TASK)
        sb = 128 if sb128 else 64
        w, h = int(geometry.pic.p[0].w), int(geometry.pic.p[0].h)
        cols = list(col_start_sb) if col_start_sb is not None else [0, (w + sb - 1) // sb]
        rows = list(row_start_sb) if row_start_sb is not None else [0, (h + sb - 1) // sb]
        ca = (C.c_uint16 * len(cols))(*cols)
        ra = (C.c_uint16 * len(rows))(*rows)
        self.h = C.c_void_p()
        _chk(ctx.lib.dav1d_hip_intra_sb_create(ctx.h, C.byref(self.h), allp.ctypes.data, ps, allt.ctypes.data, ts, len(batches),
                                               C.byref(geometry.pic), int(bool(sb128)), len(cols) - 1, ca, len(rows) - 1, ra),
             "intra_sb_create")
        self.n_levels = int(ctx.lib.dav1d_hip_intra_sb_levels(self.h))
        self.n_superblocks = int(ctx.lib.dav1d_hip_intra_sb_superblocks(self.h))

    def run(self, dst, coef, aux=None):
        _chk(self.ctx.lib.dav1d_hip_intra_sb_run(self.ctx.h, self.h, C.byref(dst.pic), coef.ptr if hasattr(coef, "ptr") else coef,
                                                 aux.ptr if aux else None), "intra_sb_run")

    def status(self):
        """dav1d_hip_intra_sb_status: waits for the run; raises when superblocks of the one-launch form were left unreconstructed"""
        n = C.c_uint32()
        _chk(self.ctx.lib.dav1d_hip_intra_sb_status(self.ctx.h, self.h, C.byref(n)), "intra_sb_status (%d superblocks gave up)" % n.value)

    def destroy(self):
        if self.h:
            self.ctx.lib.dav1d_hip_intra_sb_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class HostPictureBuf:
    """dav1d_hip_host_picture_*: pinned host planes + the device picture of the same geometry (the buffers behind a
    Dav1dPicAllocator, reference include/dav1d/picture.h)."""

    def __init__(self, ctx, w, h, layout, bpc):
        self.ctx = ctx
        self.hp = HostPicture()
        _chk(ctx.lib.dav1d_hip_host_picture_alloc(ctx.h, C.byref(self.hp), w, h, layout, bpc), "host_picture_alloc")
        self.w, self.h, self.layout, self.bpc = w, h, layout, bpc

    @property
    def dev(self):
        return self.hp.dev

    def fetch(self, src=None, row0=0, row1=1 << 30):
        _chk(self.ctx.lib.dav1d_hip_host_picture_fetch(self.ctx.h, C.byref(self.hp), C.byref(src) if src is not None else None, row0, row1),
             "host_picture_fetch")

    def wait(self):
        _chk(self.ctx.lib.dav1d_hip_host_picture_wait(self.ctx.h), "host_picture_wait")

    def plane(self, pl):
        """numpy view of host plane pl (visible w x h)."""
        p = self.hp.dev.p[pl]
        dt = np.uint16 if self.bpc > 8 else np.uint8
        stride = self.hp.stride[1 if pl else 0]
        buf = (C.c_uint8 * (stride * p.h)).from_address(self.hp.data[pl])
        return np.frombuffer(buf, dtype=dt).reshape(p.h, stride // dt().itemsize)[:, :p.w]

    def release(self):
        if self.hp.alloc:
            _chk(self.ctx.lib.dav1d_hip_host_picture_release(self.ctx.h, C.byref(self.hp)), "host_picture_release")


class FrameInFlight:
    """dav1d_hip_frame_*: one frame in flight (driver-level boundary)."""

    def __init__(self, ctx, cur, refs):
        self.ctx = ctx
        self.cur = cur
        self.h = C.c_void_p()
        arr = (Picture * max(len(refs), 1))(*[r.pic for r in refs])
        _chk(ctx.lib.dav1d_hip_frame_begin(ctx.h, C.byref(self.h), C.byref(cur.pic), arr, len(refs)), "frame_begin")

    def _cur_written(self, filtered):
        # the frame wrote cur's raster planes: its tiled twin is valid only if the frame itself retiled it (option ref_twin = 2)
        same = filtered is not None and filtered.p[0].data == self.cur.pic.p[0].data
        self.cur.pic.twin_ok = filtered.twin_ok if same else 0

    def submit_tile_sbrow(self, mc, comp, itx):
        m = np.ascontiguousarray(mc, dtype=MC_TASK)
        c = np.ascontiguousarray(comp, dtype=COMP_TASK)
        t = np.ascontiguousarray(itx, dtype=ITX_TASK)
        _chk(self.ctx.lib.dav1d_hip_frame_submit_tile_sbrow(self.h, m.ctypes.data, len(m), c.ctypes.data, len(c), t.ctypes.data, len(t)),
             "frame_submit_tile_sbrow")

    def submit_intra_step(self, step, ipred, itx, aux=None):
        a = np.ascontiguousarray(ipred, dtype=IPRED_TASK)
        t = np.ascontiguousarray(itx, dtype=ITX_TASK)
        _chk(self.ctx.lib.dav1d_hip_frame_submit_intra_step(self.h, step, a.ctypes.data, len(a), t.ctypes.data, len(t),
                                                            aux.ptr if aux is not None else None), "frame_submit_intra_step")

    def submit_filter_sbrow(self, lf, cdef, lr):
        a = np.ascontiguousarray(lf, dtype=LF_TASK)
        b = np.ascontiguousarray(cdef, dtype=CDEF_TASK)
        c = np.ascontiguousarray(lr, dtype=LR_TASK)
        _chk(self.ctx.lib.dav1d_hip_frame_submit_filter_sbrow(self.h, a.ctypes.data, len(a), b.ctypes.data, len(b), c.ctypes.data, len(c)),
             "frame_submit_filter_sbrow")

    def set_filters(self, lvl, b4_stride, lut_e, lut_i, cdef_damping, grain=None, is_id=0):
        e = np.ascontiguousarray(lut_e, dtype=np.uint8)
        i = np.ascontiguousarray(lut_i, dtype=np.uint8)
        _chk(self.ctx.lib.dav1d_hip_frame_set_filters(self.h, lvl.ptr, b4_stride, e.ctypes.data, i.ctypes.data, cdef_damping,
                                                      C.addressof(grain) if grain is not None else None, is_id), "frame_set_filters")

    def end(self, coef, prep, mask=None, grain_out=None):
        """Returns the Picture descriptor of the filtered (post CDEF / restoration) picture."""
        filtered = Picture()
        _chk(self.ctx.lib.dav1d_hip_frame_end(self.h, coef.ptr if coef is not None else None, prep.ptr if prep is not None else None,
                                              mask.ptr if mask is not None else None, C.byref(filtered),
                                              C.byref(grain_out.pic) if grain_out is not None else None), "frame_end")
        self._cur_written(filtered)
        return filtered

    def end_async(self, coef, prep, mask=None, grain_out=None, done=None):
        """dav1d_hip_frame_end_async: returns at once; done(rc) is called on the library's thread when the frame is final."""
        cb_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)
        self._cb = cb_t((lambda cookie, rc, pic: done(rc)) if done else (lambda cookie, rc, pic: None))
        _chk(self.ctx.lib.dav1d_hip_frame_end_async(self.h, coef.ptr if coef is not None else None, prep.ptr if prep is not None else None,
                                                    mask.ptr if mask is not None else None,
                                                    C.byref(grain_out.pic) if grain_out is not None else None, self._cb, None), "frame_end_async")

    def progress(self):
        return int(self.ctx.lib.dav1d_hip_frame_progress(self.h))

    def set_progress_callback(self, fn):
        """dav1d_hip_frame_set_progress_callback: fn(rows, picture) on the thread that ends the frame, every time more rows of the
        filtered picture are final (a Picture descriptor valid during the call)."""
        cb_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(Picture))
        self._pcb = cb_t(lambda cookie, rows, pic: fn(rows, pic.contents)) if fn else C.cast(None, cb_t)
        _chk(self.ctx.lib.dav1d_hip_frame_set_progress_callback(self.h, self._pcb, None), "frame_set_progress_callback")

    def wait(self):
        filtered = Picture()
        _chk(self.ctx.lib.dav1d_hip_frame_wait(self.h, C.byref(filtered)), "frame_wait")
        self._cur_written(filtered)
        return filtered

    def post_bands(self):
        """Bands the post filters of the last end() were pipelined over (0: stage by stage)."""
        return int(self.ctx.lib.dav1d_hip_frame_post_bands(self.h))

    def destroy(self):
        if self.h:
            self.ctx.lib.dav1d_hip_frame_destroy(self.h)
            self.h = C.c_void_p()
