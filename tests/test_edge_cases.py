"""Degenerate and hostile inputs at the C-ABI boundary: empty lists are no-ops, malformed tasks are refused with
-EINVAL before anything is launched, ragged picture sizes get the reference's padded geometry."""
import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth


def test_empty_batches_are_noops(ctx):
    pic = ctx.picture(64, 64, api.LAYOUT_I420, 8)
    plane = np.arange(pic.padded_shape(0)[0] * pic.padded_shape(0)[1], dtype=np.uint32).reshape(pic.padded_shape(0)).astype(np.uint8)
    pic.upload(0, plane)
    buf = ctx.buffer(64)
    ctx.itx_add_batch(pic, np.zeros(0, api.ITX_TASK), buf)
    ctx.mc_batch(pic, [pic], np.zeros(0, api.MC_TASK), buf)
    ctx.comp_batch(pic, np.zeros(0, api.COMP_TASK), buf, None)
    ctx.cdef_batch(pic, pic, np.zeros(0, api.CDEF_TASK), 3)
    ctx.lf_batch(pic, np.zeros(0, api.LF_TASK), buf, 16, np.zeros(64, np.uint8), np.zeros(64, np.uint8))
    ctx.ipred_batch(pic, np.zeros(0, api.IPRED_TASK))
    ctx.lr_batch(pic, pic, pic, np.zeros(0, api.LR_TASK))
    ctx.warp_batch(pic, [pic], np.zeros(0, api.WARP_TASK), buf)
    ctx.mc_scaled_batch(pic, [pic], np.zeros(0, api.MC_SCALED_TASK), buf)
    lst = ctx.inter_list(np.zeros(0, api.MC_TASK), np.zeros(0, api.COMP_TASK))
    ctx.run_inter_list(lst, pic, [pic], buf)
    lst.destroy()
    assert np.array_equal(pic.download(0), plane)
    pic.free(); buf.free()


@pytest.mark.parametrize("what", ["itx_tx", "itx_txtp", "mc_w", "mc_filter", "mc_ref", "comp_kind", "lr_w", "lr_type", "cdef_edges",
                                  "ipred_mode", "scaled_mx"])
def test_malformed_tasks_are_refused(ctx, what):
    pic = ctx.picture(64, 64, api.LAYOUT_I420, 8)
    buf = ctx.buffer(1 << 16)
    with pytest.raises(api.HipError) as e:
        if what.startswith("itx"):
            t = np.zeros(1, api.ITX_TASK)
            if what == "itx_tx":
                t["tx"] = 19
            else:
                t["tx"], t["txtp"] = 4, 1          # 64x64 only has DCT_DCT
            ctx.itx_add_batch(pic, t, buf)
        elif what.startswith("mc"):
            t = np.zeros(1, api.MC_TASK)
            t["w"] = t["h"] = 8
            if what == "mc_w":
                t["w"] = 3
            elif what == "mc_filter":
                t["filter_2d"] = 10
            else:
                t["ref"] = 1
            ctx.mc_batch(pic, [pic], t, buf)
        elif what == "comp_kind":
            t = np.zeros(1, api.COMP_TASK)
            t["w"] = t["h"] = 8
            t["kind"] = 7
            ctx.comp_batch(pic, t, buf, None)
        elif what.startswith("lr"):
            t = np.zeros(1, api.LR_TASK)
            t["w"], t["h"] = (385, 8) if what == "lr_w" else (8, 8)
            t["type"] = 5 if what == "lr_type" else 0
            ctx.lr_batch(pic, pic, pic, t)
        elif what == "cdef_edges":
            t = np.zeros(1, api.CDEF_TASK)
            t["edges"] = 16
            ctx.cdef_batch(pic, pic, t, 3)
        elif what == "ipred_mode":
            t = np.zeros(1, api.IPRED_TASK)
            t["tw"] = t["th"] = 1
            t["mode"] = 14
            ctx.ipred_batch(pic, t)
        else:
            t = np.zeros(1, api.MC_SCALED_TASK)
            t["w"] = t["h"] = 8
            t["mx"] = 1024
            ctx.mc_scaled_batch(pic, [pic], t, buf)
    assert "errno 22" in str(e.value), str(e.value)        # -EINVAL
    pic.free(); buf.free()


@pytest.mark.parametrize("w,h", [(1, 1), (130, 66), (1025, 9)])
def test_ragged_pictures_follow_the_reference_geometry(ctx, w, h):
    """src/picture.c:46-78: dimensions padded to 128, stride +64 bytes when it would be a multiple of 1024."""
    for bpc in (8, 10):
        pic = ctx.picture(w, h, api.LAYOUT_I420, bpc)
        geo = synth.plane_geometry(w, h, bpc, 1)
        for pl in range(3):
            assert pic.stride_px(pl) == geo[pl][0]
            assert pic.padded_shape(pl)[0] == geo[pl][1]
        a = np.random.default_rng(w).integers(0, 1 << bpc, size=pic.padded_shape(1)).astype(util.pix_dtype(bpc))
        pic.upload(1, a)
        assert np.array_equal(pic.download(1), a)
        pic.free()


def test_last_hip_error_is_reported_as_text(ctx):
    """dav1d_hip_last_hip_error: the C ABI speaks errno; the HIP error behind an -EIO / -ENOMEM is kept per thread as text."""
    import ctypes as C
    code = C.c_int(-1)
    msg = ctx.lib.dav1d_hip_last_hip_error(C.byref(code))
    assert isinstance(msg, bytes) and len(msg) > 0
    assert (code.value == 0) == (msg == b"no error")
