"""BASELINE configs[0] as a test: the reference's own pass 2 and in-loop filters run once with their C DSP functions and once with the
reference-signature table dav1d_hip_dsp_init_* fills (the kernel-level drop-in) — every mc / itx / ipred / loop filter /
CDEF / restoration call of a whole frame goes through the HIP kernels one staged call at a time.  Same picture."""
import pytest

import lister_util as lu


@pytest.mark.parametrize("bpc", [8, 10])
def test_reference_drivers_through_the_hip_dsp_table(ctx, bpc):
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (192, 136) if ctx.backend == "emu" else (640, 360)
    out = lu.c0_line(ctx, w, h, bpc)
    assert out["parity"].startswith("bit-exact") and out["cpu_c_1_thread"]["value"] > 0
