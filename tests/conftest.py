import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pt():
    """The product's ctypes view (cuda-pathtrace_amd/__init__.py -> libptcore.so)."""
    import __graft_entry__ as ge

    return ge.load_package()


@pytest.fixture(scope="session")
def lab():
    """The same view bound to libptcore_lab.so: every experimental kernel variant + the pt_debug_* diagnostics."""
    import __graft_entry__ as ge

    return ge.load_lab()


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure)."""
    import __graft_entry__ as ge

    o = ge.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def gpu(pt):
    if pt.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    pt.set_device(0)
    import __graft_entry__ as ge

    ge.load_lab().set_device(0)
    return pt.device_info()


GOLDEN = os.path.join(ROOT, "tests", "golden")
