import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from neumann_amd import _capi
        import ctypes as C
        n = C.c_int32(0)
        _capi.load().nmn_device_count(C.byref(n))
        return n.value
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu_available():
    return _gpu_count() > 0


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: only auto-skip when the marker
    # expression was not asked for explicitly.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
