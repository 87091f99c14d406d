"""ctypes binding of the C ABI in include/dav1d_hip.h.

The product library is dav1d_amd/libdav1d_hip.so (built by dav1d_amd.build / __graft_entry__.build
with hipcc for gfx950).  There is no CPU fallback: if the library is missing, or no
MI355X-class device can be opened, loading / opening raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(HERE, "libdav1d_hip.so")


class Plane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_ssize_t), ("w", C.c_int), ("h", C.c_int)]


class Picture(C.Structure):
    _fields_ = [("p", Plane * 3), ("bpc", C.c_int), ("layout", C.c_int),
                ("alloc", C.c_void_p), ("alloc_size", C.c_size_t), ("twin", C.c_void_p * 3), ("twin_alloc", C.c_void_p),
                ("twin_ok", C.c_int)]


class HostPicture(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("stride", C.c_ssize_t * 2), ("dev", Picture), ("alloc", C.c_void_p), ("alloc_size", C.c_size_t)]


# numpy mirrors of the POD task descriptors (include/dav1d_hip.h)
ITX_TASK = np.dtype([("dst_off", "<u4"), ("cf_off", "<u4"), ("eob", "<i2"), ("tx", "u1"), ("txtp", "u1"),
                     ("plane", "u1"), ("flags", "u1"), ("rsv", "u1", (2,))], align=False)
MC_TASK = np.dtype([("dst_off", "<u4"), ("src_x", "<i4"), ("src_y", "<i4"), ("w", "u1"), ("h", "u1"),
                    ("mx", "u1"), ("my", "u1"), ("filter_2d", "u1"), ("kind", "u1"), ("plane", "u1"),
                    ("ref", "u1"), ("pad", "<u4")], align=False)
COMP_TASK = np.dtype([("dst_off", "<u4"), ("tmp1_off", "<u4"), ("tmp2_off", "<u4"), ("mask_off", "<u4"),
                      ("w", "u1"), ("h", "u1"), ("kind", "u1"), ("plane", "u1"), ("arg", "i1"), ("ss", "u1"),
                      ("pad", "<u2")], align=False)
CDEF_TASK = np.dtype([("bx", "<u2"), ("by", "<u2"), ("y_pri", "u1"), ("y_sec", "u1"), ("uv_pri", "u1"), ("uv_sec", "u1"),
                      ("edges", "u1"), ("flags", "u1"), ("dir", "u1"), ("plane", "u1"), ("pad", "u1", (4,))], align=False)
LF_TASK = np.dtype([("dst_off", "<u4"), ("lvl_off", "<u4"), ("vmask", "<u4", (3,)), ("plane", "u1"), ("dir", "u1"),
                    ("lvl_comp", "u1"), ("pad", "u1")], align=False)
assert LF_TASK.itemsize == 24
IPRED_TASK = np.dtype([("dst_off", "<u4"), ("aux_off", "<u4"), ("x4", "<u2"), ("y4", "<u2"), ("w4", "<u2"), ("h4", "<u2"),
                       ("tw", "u1"), ("th", "u1"), ("mode", "u1"), ("angle", "i1"), ("flags", "u1"), ("plane", "u1"),
                       ("kind", "u1"), ("pad", "u1"), ("max_w", "<u2"), ("max_h", "<u2"), ("pal", "<u2", (8,))], align=False)
assert IPRED_TASK.itemsize == 44
LR_TASK = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "<u2"), ("h", "<u2"), ("plane", "u1"), ("edges", "u1"), ("type", "u1"),
                    ("pad", "u1"), ("filter", "<i2", (2, 8))], align=False)
assert LR_TASK.itemsize == 44
WARP_TASK = np.dtype([("dst_off", "<u4"), ("src_x", "<i4"), ("src_y", "<i4"), ("mx", "<i4"), ("my", "<i4"), ("abcd", "<i2", (4,)),
                      ("tmp_stride", "<u2"), ("kind", "u1"), ("plane", "u1"), ("ref", "u1"), ("pad", "u1", (3,))], align=False)
assert WARP_TASK.itemsize == 36
MC_SCALED_TASK = np.dtype([("dst_off", "<u4"), ("src_x", "<i4"), ("src_y", "<i4"), ("mx", "<i2"), ("my", "<i2"), ("dx", "<i2"),
                           ("dy", "<i2"), ("w", "u1"), ("h", "u1"), ("filter_2d", "u1"), ("kind", "u1"), ("plane", "u1"),
                           ("ref", "u1"), ("pad", "u1", (2,))], align=False)
assert MC_SCALED_TASK.itemsize == 28
assert CDEF_TASK.itemsize == 16
assert ITX_TASK.itemsize == 16 and MC_TASK.itemsize == 24 and COMP_TASK.itemsize == 24

# every symbol include/dav1d_hip.h declares (tests check the built library exports all of them)
SYMBOLS = [
    "dav1d_hip_open", "dav1d_hip_set_option", "dav1d_hip_get_option", "dav1d_hip_close", "dav1d_hip_sync", "dav1d_hip_stream", "dav1d_hip_recon_list_create", "dav1d_hip_recon_list_run", "dav1d_hip_recon_list_run_twin", "dav1d_hip_recon_list_run_tiled", "dav1d_hip_recon_list_run_tiled_timed", "dav1d_hip_picture_untile", "dav1d_hip_recon_list_destroy", "dav1d_hip_recon_list_run_timed", "dav1d_hip_fg_prepare", "dav1d_hip_fg_apply_prepared", "dav1d_hip_fg_grain_destroy", "dav1d_hip_intra_list_create", "dav1d_hip_intra_list_run_batch", "dav1d_hip_intra_list_run_all", "dav1d_hip_intra_list_destroy", "dav1d_hip_intra_flow_create", "dav1d_hip_intra_flow_run", "dav1d_hip_intra_flow_destroy", "dav1d_hip_intra_flow_units", "dav1d_hip_intra_flow_status", "dav1d_hip_intra_sb_create", "dav1d_hip_intra_sb_run", "dav1d_hip_intra_sb_status", "dav1d_hip_intra_sb_destroy", "dav1d_hip_intra_sb_levels", "dav1d_hip_intra_sb_superblocks", "dav1d_hip_frame_set_tiling", "dav1d_hip_frame_submit_intra_step", "dav1d_hip_frame_submit_intra_sorted", "dav1d_hip_frame_submit_step_copy", "dav1d_hip_frame_flush", "dav1d_hip_frame_set_refs", "dav1d_hip_frame_set_super_res", "dav1d_hip_frame_end_async", "dav1d_hip_frame_progress", "dav1d_hip_frame_wait", "dav1d_hip_frame_set_progress_callback", "dav1d_hip_host_picture_alloc", "dav1d_hip_host_picture_release", "dav1d_hip_host_picture_fetch", "dav1d_hip_host_picture_wait", "dav1d_hip_lf_rects", "dav1d_hip_lf_rects_sb", "dav1d_hip_lf_rects_free", "dav1d_hip_lf_masks_build", "dav1d_hip_refmvs_splat_batch", "dav1d_hip_refmvs_save_tmvs", "dav1d_hip_frame_post_bands", "dav1d_hip_graph_begin", "dav1d_hip_graph_end", "dav1d_hip_graph_launch", "dav1d_hip_graph_destroy", "dav1d_hip_graph_nodes", "dav1d_hip_version", "dav1d_hip_last_kernel_ms", "dav1d_hip_last_hip_error",
    "dav1d_hip_malloc", "dav1d_hip_free", "dav1d_hip_memset", "dav1d_hip_upload", "dav1d_hip_download",
    "dav1d_hip_live_objects", "dav1d_hip_device_count", "dav1d_hip_context_device", "dav1d_hip_context_use", "dav1d_hip_current_device", "dav1d_hip_set_device", "dav1d_hip_picture_device", "dav1d_hip_picture_copy_peer", "dav1d_hip_picture_copy_peer_rows", "dav1d_hip_enable_peer_access", "dav1d_hip_picture_alloc", "dav1d_hip_picture_free", "dav1d_hip_picture_twin_alloc", "dav1d_hip_picture_retile", "dav1d_hip_picture_retile_overlapped", "dav1d_hip_peer_unique_id", "dav1d_hip_peer_open", "dav1d_hip_peer_close", "dav1d_hip_peer_rank", "dav1d_hip_peer_world", "dav1d_hip_peer_broadcast_picture", "dav1d_hip_peer_allgather_columns", "dav1d_hip_peer_exchange_halo", "dav1d_hip_peer_allgather_columns_async", "dav1d_hip_peer_wait", "dav1d_hip_plane_upload", "dav1d_hip_plane_download",
    "dav1d_hip_itx_add_batch", "dav1d_hip_itx_list_create", "dav1d_hip_itx_list_destroy", "dav1d_hip_itx_list_run",
    "dav1d_hip_mc_batch", "dav1d_hip_mc_list_create", "dav1d_hip_mc_list_destroy", "dav1d_hip_mc_list_run",
    "dav1d_hip_comp_batch", "dav1d_hip_comp_list_create", "dav1d_hip_comp_list_destroy", "dav1d_hip_comp_list_run",
    "dav1d_hip_dsp_init_8bpc", "dav1d_hip_dsp_init_16bpc",
    "dav1d_hip_itx_list_run_timed", "dav1d_hip_mc_list_run_timed",
    "dav1d_hip_inter_list_create", "dav1d_hip_inter_list_destroy", "dav1d_hip_inter_list_run",
    "dav1d_hip_inter_list_run_timed", "dav1d_hip_inter_list_fused",
    "dav1d_hip_cdef_batch", "dav1d_hip_lf_batch", "dav1d_hip_ipred_batch", "dav1d_hip_lr_batch",
    "dav1d_hip_fg_apply", "dav1d_hip_fg_generate_grain",
    "dav1d_hip_warp_batch", "dav1d_hip_mc_scaled_batch", "dav1d_hip_resize", "dav1d_hip_emu_edge",
    "dav1d_hip_ipred_list_create", "dav1d_hip_ipred_list_run_batch", "dav1d_hip_ipred_list_destroy",
    "dav1d_hip_frame_begin", "dav1d_hip_frame_submit_tile_sbrow", "dav1d_hip_frame_submit_coefs", "dav1d_hip_frame_coef_bytes", "dav1d_hip_frame_submit_filter_sbrow",
    "dav1d_hip_frame_set_filters", "dav1d_hip_frame_end", "dav1d_hip_frame_destroy",
    "dav1d_hip_frame_submit_step_blend", "dav1d_hip_frame_submit_warp", "dav1d_hip_frame_submit_scaled",
    "dav1d_hip_lister_create", "dav1d_hip_lister_tile_sbrow", "dav1d_hip_lister_run", "dav1d_hip_lister_filter_run", "dav1d_hip_lister_run_frame", "dav1d_hip_lister_prep_elems", "dav1d_hip_lister_mask_bytes",
    "dav1d_hip_lister_steps", "dav1d_hip_lister_const_masks", "dav1d_hip_lister_destroy",
    "dav1d_hip_lister_mask_offset", "dav1d_hip_lister_tables", "dav1d_hip_lister_block_warp", "dav1d_hip_lister_filter_sbrow",
]


class FilmGrainData(C.Structure):      # == Dav1dFilmGrainData, reference include/dav1d/headers.h:315-333
    _fields_ = [("seed", C.c_uint), ("num_y_points", C.c_int), ("y_points", (C.c_uint8 * 2) * 14),
                ("chroma_scaling_from_luma", C.c_int), ("num_uv_points", C.c_int * 2), ("uv_points", ((C.c_uint8 * 2) * 10) * 2),
                ("scaling_shift", C.c_int), ("ar_coeff_lag", C.c_int), ("ar_coeffs_y", C.c_int8 * 24),
                ("ar_coeffs_uv", (C.c_int8 * 28) * 2), ("ar_coeff_shift", C.c_uint64), ("grain_scale_shift", C.c_int),
                ("uv_mult", C.c_int * 2), ("uv_luma_mult", C.c_int * 2), ("uv_offset", C.c_int * 2),
                ("overlap_flag", C.c_int), ("clip_to_restricted_range", C.c_int)]


class WarpParams(C.Structure):         # == Dav1dHipWarpParams / Dav1dWarpedMotionParams
    _fields_ = [("type", C.c_int), ("matrix", C.c_int32 * 6), ("abcd", C.c_int16 * 4)]


class FrameDesc(C.Structure):          # == Dav1dHipFrameDesc
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("layout", C.c_int), ("bpc", C.c_int), ("sb128", C.c_int),
                ("intra_edge_filter", C.c_int), ("is_inter", C.c_int), ("n_tile_cols", C.c_int), ("n_tile_rows", C.c_int),
                ("col_start_sb", C.c_uint16 * 65), ("row_start_sb", C.c_uint16 * 65), ("b4_stride", C.c_ssize_t),
                ("b", C.c_void_p), ("cbi", C.c_void_p), ("tile_start_off", C.c_void_p), ("pal", C.c_void_p),
                ("svc", ((C.c_int32 * 2) * 2) * 7), ("ref_w", C.c_int * 7), ("ref_h", C.c_int * 7), ("gmv", WarpParams * 7),
                ("gmv_warp_allowed", C.c_uint8 * 7), ("jnt_weights", (C.c_uint8 * 7) * 7), ("cf_align64", C.c_int),
                ("lossless", C.c_uint8 * 8), ("cf", C.c_void_p)]


class FilterDesc(C.Structure):         # == Dav1dHipFilterDesc
    _fields_ = [("lf_level_y", C.c_int * 2), ("lf_level_u", C.c_int), ("lf_level_v", C.c_int), ("lf_mask", C.c_void_p),
                ("tx_lpf_right_edge", C.c_void_p * 2), ("a_tx_lpf_y", C.c_void_p), ("a_tx_lpf_uv", C.c_void_p), ("a_stride", C.c_size_t),
                ("cdef_enabled", C.c_int), ("cdef_damping", C.c_int), ("cdef_y_strength", C.c_uint8 * 8), ("cdef_uv_strength", C.c_uint8 * 8),
                ("lr_type", C.c_int * 3), ("lr_unit_size", C.c_int * 2), ("lr_mask", C.c_void_p), ("sr_w", C.c_int)]


class LibraryError(RuntimeError):
    pass


def load(path=None):
    """dlopen the C-ABI library and declare prototypes.  Raises if it is absent."""
    path = path or DEFAULT_PATH
    if not os.path.exists(path):
        raise LibraryError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    P = C.POINTER
    protos = {
        "dav1d_hip_open": (i, [P(vp), i, vp]),
        "dav1d_hip_close": (None, [vp]),
        "dav1d_hip_sync": (i, [vp]),
        "dav1d_hip_stream": (vp, [vp]),
        "dav1d_hip_recon_list_create": (i, [vp, vp, vp, vp, sz, vp, sz, vp, sz]),
        "dav1d_hip_recon_list_run": (i, [vp, vp, vp, vp, i, vp, vp, vp]),
        "dav1d_hip_recon_list_run_twin": (i, [vp, vp, vp, vp, i, vp, vp, vp]),
        "dav1d_hip_recon_list_run_tiled": (i, [vp, vp, vp, vp, i, vp, vp, vp]),
        "dav1d_hip_recon_list_destroy": (None, [vp, vp]),
        "dav1d_hip_recon_list_run_timed": (i, [vp, vp, vp, vp, i, vp, vp, vp, vp, vp]),
        "dav1d_hip_recon_list_run_tiled_timed": (i, [vp, vp, vp, vp, i, vp, vp, vp, vp, vp]),
        "dav1d_hip_fg_prepare": (i, [vp, vp, vp, i, i]),
        "dav1d_hip_fg_apply_prepared": (i, [vp, vp, vp, vp, i]),
        "dav1d_hip_fg_grain_destroy": (None, [vp, vp]),
        "dav1d_hip_intra_list_create": (i, [vp, vp, vp, vp, vp, vp, sz]),
        "dav1d_hip_intra_list_run_batch": (i, [vp, vp, sz, vp, vp, vp]),
        "dav1d_hip_intra_list_run_all": (i, [vp, vp, vp, vp, vp]),
        "dav1d_hip_intra_list_destroy": (None, [vp, vp]),
        "dav1d_hip_set_option": (i, [vp, C.c_char_p, C.c_long]),
        "dav1d_hip_get_option": (i, [vp, C.c_char_p, C.POINTER(C.c_long)]),
        "dav1d_hip_intra_flow_create": (i, [vp, vp, vp, vp, vp, vp, sz]),
        "dav1d_hip_intra_flow_run": (i, [vp, vp, vp, vp, vp]),
        "dav1d_hip_intra_flow_destroy": (None, [vp, vp]),
        "dav1d_hip_intra_flow_units": (sz, [vp]),
        "dav1d_hip_intra_flow_status": (i, [vp, vp, vp]),
        "dav1d_hip_intra_sb_create": (i, [vp, vp, vp, vp, vp, vp, sz, vp, i, i, vp, i, vp]),
        "dav1d_hip_intra_sb_run": (i, [vp, vp, vp, vp, vp]),
        "dav1d_hip_intra_sb_status": (i, [vp, vp, vp]),
        "dav1d_hip_intra_sb_destroy": (None, [vp, vp]),
        "dav1d_hip_intra_sb_levels": (sz, [vp]),
        "dav1d_hip_intra_sb_superblocks": (sz, [vp]),
        "dav1d_hip_frame_set_tiling": (i, [vp, i, i, vp, i, vp]),
        "dav1d_hip_frame_submit_intra_step": (i, [vp, sz, vp, sz, vp, sz, vp]),
        "dav1d_hip_frame_submit_intra_sorted": (i, [vp, sz, vp, vp, vp, vp, vp, vp]),
        "dav1d_hip_frame_end_async": (i, [vp, vp, vp, vp, vp, vp, vp]),
        "dav1d_hip_frame_progress": (i, [vp]),
        "dav1d_hip_frame_wait": (i, [vp, vp]),
        "dav1d_hip_frame_submit_step_copy": (i, [vp, vp, vp, sz]),
        "dav1d_hip_frame_set_super_res": (i, [vp, i]),
        "dav1d_hip_frame_flush": (i, [vp]),
        "dav1d_hip_frame_set_refs": (i, [vp, P(Picture), i]),
        "dav1d_hip_frame_set_progress_callback": (i, [vp, vp, vp]),
        "dav1d_hip_host_picture_alloc": (i, [vp, vp, i, i, i, i]),
        "dav1d_hip_host_picture_release": (i, [vp, vp]),
        "dav1d_hip_host_picture_fetch": (i, [vp, vp, vp, i, i]),
        "dav1d_hip_host_picture_wait": (i, [vp]),
        "dav1d_hip_lf_rects": (i, [vp, vp, vp, vp]),
        "dav1d_hip_lf_rects_sb": (i, [vp, vp, vp, vp, vp]),
        "dav1d_hip_lf_rects_free": (None, [vp]),
        "dav1d_hip_lf_masks_build": (i, [vp, vp, vp, sz, vp, vp, vp, vp, vp]),
        "dav1d_hip_refmvs_splat_batch": (i, [vp, vp, C.c_ssize_t, vp, sz]),
        "dav1d_hip_refmvs_save_tmvs": (i, [vp, vp, C.c_ssize_t, vp, C.c_ssize_t, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "dav1d_hip_frame_post_bands": (i, [vp]),
        "dav1d_hip_graph_begin": (i, [vp]),
        "dav1d_hip_graph_end": (i, [vp, vp]),
        "dav1d_hip_graph_launch": (i, [vp, vp]),
        "dav1d_hip_graph_destroy": (None, [vp, vp]),
        "dav1d_hip_graph_nodes": (sz, [vp]),
        "dav1d_hip_version": (C.c_char_p, []),
        "dav1d_hip_last_hip_error": (C.c_char_p, [vp]),
        "dav1d_hip_last_kernel_ms": (C.c_float, [vp]),
        "dav1d_hip_malloc": (i, [vp, P(vp), sz]),
        "dav1d_hip_free": (i, [vp, vp]),
        "dav1d_hip_memset": (i, [vp, vp, i, sz]),
        "dav1d_hip_upload": (i, [vp, vp, vp, sz]),
        "dav1d_hip_download": (i, [vp, vp, vp, sz]),
        "dav1d_hip_picture_alloc": (i, [vp, P(Picture), i, i, i, i]),
        "dav1d_hip_picture_free": (i, [vp, P(Picture)]),
        "dav1d_hip_picture_twin_alloc": (i, [vp, P(Picture)]),
        "dav1d_hip_picture_untile": (i, [vp, P(Picture)]),
        "dav1d_hip_picture_retile": (i, [vp, P(Picture)]),
        "dav1d_hip_picture_retile_overlapped": (i, [vp, P(Picture)]),
        "dav1d_hip_peer_unique_id": (i, [vp]),
        "dav1d_hip_peer_open": (i, [vp, P(vp), vp, i, i]),
        "dav1d_hip_peer_close": (None, [vp]),
        "dav1d_hip_peer_rank": (i, [vp]),
        "dav1d_hip_peer_world": (i, [vp]),
        "dav1d_hip_peer_broadcast_picture": (i, [vp, P(Picture), i]),
        "dav1d_hip_peer_allgather_columns": (i, [vp, P(Picture), vp, vp]),
        "dav1d_hip_peer_exchange_halo": (i, [vp, P(Picture), vp, vp, i]),
        "dav1d_hip_peer_allgather_columns_async": (i, [vp, P(Picture), vp, vp]),
        "dav1d_hip_peer_wait": (i, [vp, i]),
        "dav1d_hip_plane_upload": (i, [vp, P(Picture), i, vp, C.c_ssize_t, i]),
        "dav1d_hip_plane_download": (i, [vp, P(Picture), i, vp, C.c_ssize_t, i]),
        "dav1d_hip_itx_add_batch": (i, [vp, P(Picture), vp, sz, vp]),
        "dav1d_hip_itx_list_create": (i, [vp, P(vp), vp, sz]),
        "dav1d_hip_itx_list_destroy": (None, [vp, vp]),
        "dav1d_hip_itx_list_run": (i, [vp, vp, P(Picture), vp]),
        "dav1d_hip_mc_batch": (i, [vp, P(Picture), P(Picture), i, vp, sz, vp]),
        "dav1d_hip_mc_list_create": (i, [vp, P(vp), vp, sz]),
        "dav1d_hip_mc_list_destroy": (None, [vp, vp]),
        "dav1d_hip_mc_list_run": (i, [vp, vp, P(Picture), P(Picture), i, vp]),
        "dav1d_hip_comp_batch": (i, [vp, P(Picture), vp, sz, vp, vp]),
        "dav1d_hip_comp_list_create": (i, [vp, P(vp), vp, sz]),
        "dav1d_hip_comp_list_destroy": (None, [vp, vp]),
        "dav1d_hip_comp_list_run": (i, [vp, vp, P(Picture), vp, vp]),
        "dav1d_hip_itx_list_run_timed": (i, [vp, vp, P(Picture), vp, P(C.c_float), P(sz)]),
        "dav1d_hip_mc_list_run_timed": (i, [vp, vp, P(Picture), P(Picture), i, vp, P(C.c_float), P(sz)]),
        "dav1d_hip_inter_list_create": (i, [vp, P(vp), vp, sz, vp, sz]),
        "dav1d_hip_inter_list_destroy": (None, [vp, vp]),
        "dav1d_hip_inter_list_run": (i, [vp, vp, P(Picture), P(Picture), i, vp, vp]),
        "dav1d_hip_inter_list_run_timed": (i, [vp, vp, P(Picture), P(Picture), i, vp, vp, P(C.c_float), P(sz)]),
        "dav1d_hip_inter_list_fused": (sz, [vp]),
        "dav1d_hip_cdef_batch": (i, [vp, P(Picture), P(Picture), vp, sz, i, vp]),
        "dav1d_hip_lf_batch": (i, [vp, P(Picture), vp, sz, vp, C.c_ssize_t, vp, vp]),
        "dav1d_hip_ipred_batch": (i, [vp, P(Picture), vp, sz, vp]),
        "dav1d_hip_lr_batch": (i, [vp, P(Picture), P(Picture), P(Picture), vp, sz]),
        "dav1d_hip_fg_apply": (i, [vp, P(Picture), P(Picture), P(FilmGrainData), i]),
        "dav1d_hip_fg_generate_grain": (i, [vp, P(FilmGrainData), i, i, vp]),
        "dav1d_hip_warp_batch": (i, [vp, P(Picture), P(Picture), i, vp, sz, vp]),
        "dav1d_hip_mc_scaled_batch": (i, [vp, P(Picture), P(Picture), i, vp, sz, vp]),
        "dav1d_hip_resize": (i, [vp, P(Picture), P(Picture), i, i, i, i, i, i, i]),
        "dav1d_hip_emu_edge": (i, [vp, i, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t, vp,
                                   C.c_ssize_t, vp, C.c_ssize_t]),
        "dav1d_hip_ipred_list_create": (i, [vp, P(vp), vp, vp, sz]),
        "dav1d_hip_ipred_list_run_batch": (i, [vp, vp, sz, P(Picture), vp]),
        "dav1d_hip_ipred_list_destroy": (None, [vp, vp]),
        "dav1d_hip_frame_begin": (i, [vp, P(vp), P(Picture), P(Picture), i]),
        "dav1d_hip_frame_submit_tile_sbrow": (i, [vp, vp, sz, vp, sz, vp, sz]),
        "dav1d_hip_frame_submit_coefs": (i, [vp, vp, sz, vp]),
        "dav1d_hip_frame_coef_bytes": (sz, [vp]),
        "dav1d_hip_frame_submit_filter_sbrow": (i, [vp, vp, sz, vp, sz, vp, sz]),
        "dav1d_hip_frame_set_filters": (i, [vp, vp, C.c_ssize_t, vp, vp, i, vp, i]),
        "dav1d_hip_frame_end": (i, [vp, vp, vp, vp, P(Picture), P(Picture)]),
        "dav1d_hip_frame_destroy": (None, [vp]),
        "dav1d_hip_frame_submit_step_blend": (i, [vp, sz, vp, sz]),
        "dav1d_hip_frame_submit_warp": (i, [vp, vp, sz]),
        "dav1d_hip_frame_submit_scaled": (i, [vp, vp, sz]),
        "dav1d_hip_lister_create": (i, [P(vp), P(FrameDesc), vp]),
        "dav1d_hip_lister_tile_sbrow": (i, [vp, i, i, i]),
        "dav1d_hip_lister_run": (i, [vp, i]),
        "dav1d_hip_lister_filter_run": (i, [vp, vp, i]),
        "dav1d_hip_lister_run_frame": (i, [vp, vp, i]),
        "dav1d_hip_lister_prep_elems": (sz, [vp]),
        "dav1d_hip_lister_mask_bytes": (sz, [vp]),
        "dav1d_hip_lister_steps": (sz, [vp]),
        "dav1d_hip_lister_const_masks": (vp, [P(sz)]),
        "dav1d_hip_lister_destroy": (None, [vp]),
        "dav1d_hip_lister_mask_offset": (C.c_long, [i, i, i, i, i]),
        "dav1d_hip_lister_tables": (None, [vp]),
        "dav1d_hip_lister_filter_sbrow": (i, [vp, P(FilterDesc), i]),
        "dav1d_hip_lister_block_warp": (i, [P(WarpParams), vp, vp, i, i, i, i]),
        "dav1d_hip_device_count": (i, []),
        "dav1d_hip_context_device": (i, [vp]),
        "dav1d_hip_context_use": (i, [vp]),
        "dav1d_hip_current_device": (i, []),
        "dav1d_hip_set_device": (i, [i]),
        "dav1d_hip_picture_device": (i, [P(Picture)]),
        "dav1d_hip_picture_copy_peer": (i, [vp, P(Picture), vp, P(Picture)]),
        "dav1d_hip_picture_copy_peer_rows": (i, [vp, P(Picture), vp, P(Picture), C.c_int, C.c_int]),
        "dav1d_hip_enable_peer_access": (i, [vp, vp]),
        "dav1d_hip_dsp_init_8bpc": (i, [vp]),
        "dav1d_hip_dsp_init_16bpc": (i, [vp, i]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
