import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def testscan():
    from libwave_amd.pcd import load_pcd_xyz
    return load_pcd_xyz(os.path.join(ROOT, "tests", "golden", "testscan.pcd"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def wm():
    """The product's C ABI (requires the in-tree libwavematch_hip.so).  No CPU
    fallback exists: on a box without a HIP device Context() raises."""
    import __graft_entry__ as g
    g.build()
    from libwave_amd import capi
    capi.lib()
    return capi


@pytest.fixture()
def ctx(wm):
    c = wm.Context(0)
    yield c
    c.close()
