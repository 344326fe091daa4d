import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    """a HIP device AND the built library: without either, `-m gpu` tests are skipped instead of killing the session
    (ggml_hip_init exits the process when no device is visible -- the backend has no CPU fallback by design)"""
    lib = os.path.join(ROOT, "ggllm.cpp_amd", "libggml_hip.so")
    if not os.path.exists(lib):
        return False, "libggml_hip.so is not built"
    try:
        import ctypes
        L = ctypes.CDLL(lib)
        L.ggml_hip_device_count.restype = ctypes.c_int
        n = L.ggml_hip_device_count()
    except Exception as e:      # noqa: BLE001
        return False, f"libggml_hip.so does not load: {e}"
    return (n > 0), "no HIP device visible"


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items:
        return
    ok, why = _gpu_available()
    if ok:
        return
    skip = pytest.mark.skip(reason=f"gpu test skipped: {why}")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding as ob
    ob.build_oracle()
    return ob.Oracle()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return G()
