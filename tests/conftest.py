import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# The SIMT emulator shows two devices (tests/emu/emu_rt.cpp): every allocation is tagged with the device it was made on, the library's
# checks that a frame's pictures live on its context's device have something to check in every emulated test, and the binding's
# n_devices = 2 path (tests/test_hooked.py, tests/test_stream.py) runs on this box.  Read once, when the emulated library first asks.
os.environ.setdefault("DAV1D_EMU_DEVICES", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Every kernel parity test runs against two builds of the SAME kernel sources:
#   emu -> tests/emu SIMT emulation compiled with g++ (CPU container, -m "not gpu")
#   hip -> dav1d_amd/libdav1d_hip.so on cuda:0 through the C ABI (-m gpu)
BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", marks=pytest.mark.gpu, id="hip")]


@pytest.fixture(scope="session", params=BACKENDS)
def ctx(request):
    import util
    c = util.make_context(request.param)
    c.backend = request.param
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _default_options(request):
    """Tests that turn a context's tuning knobs (ctx.set_option) get the library's defaults back afterwards: the context is shared."""
    yield
    if "ctx" in request.fixturenames:
        c = request.getfixturevalue("ctx")
        for name, value in (("recon_fuse", 15), ("recon_pair_streams", 2), ("recon_pipeline", 16384), ("recon_lanes", 1), ("recon_coop_below", 4096), ("chunk_upload", 0), ("post_bands", 0), ("chunk_order", 0), ("chunk_hints", 1)):
            c.set_option(name, value)


@pytest.fixture(params=[False, True, "native"], ids=["raster-refs", "tiled-refs", "tiled-native"])
def twin_refs(request, ctx):
    """Tests that name this fixture run three times: with reference pictures read in raster order; with every uploaded picture
    retiled (api.DevicePicture.upload -> dav1d_hip_picture_retile) so that motion compensation reads references through their tiled
    twins (Dav1dHipPicture.twin, the TILED kernel variants of mc.hip / recon.hip); and "tiled-native": tiled references AND every
    reconstruction leaving its picture in the twin ONLY (dav1d_hip_recon_list_run_tiled for recon lists, context option ref_twin = 3
    for frames: DAV1D_HIP_TWIN_ONLY) — the raster planes the tests compare then come from the un-tiling download / fetch.  All three
    must give the oracle's pixels."""
    ctx.auto_retile = bool(request.param)
    ctx.tiled_native = request.param == "native"
    if ctx.tiled_native:
        ctx.set_option("ref_twin", 3)
    yield request.param
    ctx.auto_retile = False
    ctx.tiled_native = False
    ctx.set_option("ref_twin", 1)
