import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge

    p = ge.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    # GPU tests must run the HIP path or fail loudly: no skip-on-missing-device here.
    assert pkg.device_count() >= 1, "no HIP device visible: -m gpu tests need the MI355X"
    return pkg


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
