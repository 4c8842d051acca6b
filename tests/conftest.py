import os
import sys

import pytest

# the library's self-check of the inertial term's segment-wise clearing of its Hessian (ba_internal.hpp, InertialTerm::cost): on in every test run
os.environ.setdefault("TC2LI_TEST_HI_CLEAR", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes mirror of the C ABI); the native library is built on demand."""
    import __graft_entry__ as ge
    ge.build_native()
    import tc2li_loader
    return tc2li_loader.load()


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle -- the checker.  Only tests, smoke() and bench.py's cpu_baseline leg may touch it."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def synthetic(pkg):
    from tc2li_slam_amd import synthetic as s
    return s


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
