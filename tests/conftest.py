import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
        for name, value in (("recon_fuse", 14), ("recon_pipeline", 16384), ("recon_lanes", 1), ("recon_coop_below", 4096), ("chunk_upload", 0), ("post_bands", 0)):
            c.set_option(name, value)
