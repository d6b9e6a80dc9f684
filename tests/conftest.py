import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _has_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


@pytest.fixture(scope="session")
def oracle():
    import refcpu_py
    refcpu_py.lib()
    return refcpu_py


@pytest.fixture(scope="session")
def vgs():
    import vgs_svgs_segmentation_amd as v
    return v


@pytest.fixture(scope="session")
def gpu(vgs):
    if not _has_gpu():
        pytest.fail("this test is marked gpu but no HIP device is visible (the product has no CPU path)")
    return vgs
