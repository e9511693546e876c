import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/vr_oracle.c), built on demand.  Test infrastructure only."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def product_lib():
    """libvrterrain.so, built on demand (hipcc cross-compiles without a GPU)."""
    from vrenderer_amd import build as vr_build
    vr_build.build()
    import vrenderer_amd
    return vrenderer_amd.load_library()


@pytest.fixture(scope="session")
def gpu_ctx(product_lib):
    import vrenderer_amd as vr
    ctx = vr.Context(0)
    yield ctx
    ctx.close()
