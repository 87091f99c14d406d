"""Film grain parity: dav1d_hip_fg_apply (templates + scaling LUTs + per-block application on the
device) vs the reference's dav1d_apply_grain run through oracle/ref_shim.c; grain templates are also
compared on their own.  Parameter ranges follow tests/checkasm/filmgrain.c:49-390."""
import ctypes as C

import numpy as np
import pytest

import util
from dav1d_amd import api
import synth_frames as synth
from dav1d_amd._lib import FilmGrainData


def random_fg(rng, bpc, variant):
    d = FilmGrainData()
    d.seed = int(rng.integers(0, 1 << 16))
    d.num_y_points = int(rng.integers(1, 15)) if variant != 3 else 0
    xs = np.sort(rng.choice(256, size=14, replace=False))
    for i in range(14):
        d.y_points[i][0], d.y_points[i][1] = int(xs[i]), int(rng.integers(0, 256))
    d.chroma_scaling_from_luma = 1 if variant == 1 else 0
    for pl in range(2):
        d.num_uv_points[pl] = 0 if variant in (1, 2) and pl == 1 else int(rng.integers(1, 11))
        if variant == 1:
            d.num_uv_points[pl] = 0
        xs = np.sort(rng.choice(256, size=10, replace=False))
        for i in range(10):
            d.uv_points[pl][i][0], d.uv_points[pl][i][1] = int(xs[i]), int(rng.integers(0, 256))
        d.uv_mult[pl] = int(rng.integers(-128, 128))
        d.uv_luma_mult[pl] = int(rng.integers(-128, 128))
        d.uv_offset[pl] = int(rng.integers(-256, 256))
        for i in range(25):
            d.ar_coeffs_uv[pl][i] = int(rng.integers(-128, 128)) >> 2
    d.scaling_shift = int(rng.integers(8, 12))
    d.ar_coeff_lag = int(rng.integers(0, 4))
    for i in range(24):
        d.ar_coeffs_y[i] = int(rng.integers(-128, 128)) >> 2
    d.ar_coeff_shift = int(rng.integers(6, 10))
    d.grain_scale_shift = int(rng.integers(0, 4))
    d.overlap_flag = int(rng.integers(0, 2)) if variant else 1
    d.clip_to_restricted_range = int(rng.integers(0, 2))
    return d


class _Driver:
    """dav1d_apply_grain / the template generator of an oracle: the reference's own (oracle/ref_shim.c) or the
    restatement's (oracle/port/fg_port.c), same arguments."""

    def __init__(self, oracle):
        pre = "dav1d_ref_" if oracle.which == "ref" else "dav1d_port_"
        self.apply_grain = getattr(oracle.lib, pre + "apply_grain")
        self.apply_grain.restype = C.c_int
        self.apply_grain.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t]
        self.generate_grain = getattr(oracle.lib, pre + "generate_grain")
        self.generate_grain.restype = C.c_int
        self.generate_grain.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]


def fg_driver(oracle=None):
    return _Driver(oracle or util.default_oracle())


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("bpc", [8, 10, 12])
def test_film_grain_matches_reference(ctx, bpc, variant):
    lib = fg_driver()
    rng = np.random.default_rng(3000 + 10 * bpc + variant)
    layout = api.LAYOUT_I420 if variant != 2 else api.LAYOUT_I444
    w, h = (160, 96) if ctx.backend == "emu" else (736, 416)
    if variant == 3:
        w -= 1          # odd width: luma padding pixel for the chroma average
    data = random_fg(rng, bpc, variant)
    # templates
    want_lut = np.zeros((3, 74, 82), np.int16)
    lib.generate_grain(bpc, C.byref(data), layout, want_lut.ctypes.data)
    got_lut = ctx.fg_generate_grain(data, bpc, layout)
    assert np.array_equal(got_lut[0], want_lut[0]), "luma grain template"
    assert np.array_equal(got_lut, want_lut), "chroma grain templates"
    # application
    src = ctx.picture(w, h, layout, bpc)
    dst = ctx.picture(w, h, layout, bpc)
    planes = synth.make_planes(rng, w, h, bpc, smooth=True) if layout == 1 else None
    if planes is None:
        ph, pw = src.padded_shape(0)
        planes = []
        for pl in range(3):
            base = np.zeros((ph, src.stride_px(pl)), src.dtype)
            base[:, :pw] = rng.integers(0, 1 << bpc, size=(ph, pw))
            planes.append(base[:, :pw])
    out0 = [np.zeros_like(p.base) [:, :p.shape[1]] for p in planes]
    for pl in range(3):
        src.upload(pl, planes[pl])
        dst.upload(pl, out0[pl])
    want = synth.copy_planes(out0)
    inp = synth.copy_planes(planes)
    outp = (C.c_void_p * 3)(*[p.ctypes.data for p in want])
    inpp = (C.c_void_p * 3)(*[p.ctypes.data for p in inp])
    lib.apply_grain(bpc, C.byref(data), w, h, layout, int(variant == 2), outp, inpp, want[0].strides[0], want[1].strides[0])
    ss_v, ss_h = (1 if layout == 1 else 0), (1 if layout != 3 else 0)
    # one call (dav1d_apply_grain), then the two halves (prepare at "frame start", apply at the end)
    for split in (False, True):
        for pl in range(3):
            dst.upload(pl, out0[pl])
        if split:
            g = ctx.fg_prepare(data, bpc, layout)
            ctx.fg_apply_prepared(dst, src, g, int(variant == 2))
            ctx.fg_grain_destroy(g)
        else:
            ctx.fg_apply(dst, src, data, int(variant == 2))
        for pl in range(3):
            vh, vw = (h, w) if pl == 0 else ((h + ss_v) >> ss_v, (w + ss_h) >> ss_h)
            got = dst.download(pl)[:vh, :vw]
            bad = np.argwhere(got != want[pl][:vh, :vw])
            assert not len(bad), "plane %d differs at %s: got %d want %d (%d px, split=%s)" % (
                pl, bad[0], got[tuple(bad[0])], want[pl][tuple(bad[0])], len(bad), split)
    src.free(); dst.free()
