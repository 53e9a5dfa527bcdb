import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "accelerated-lpbox-admm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _build_if_stale()


def _build_if_stale():
    """A fresh checkout has no liblpbox_hip.so (it is git-ignored): build it, as __graft_entry__.build() would.  Nothing is rebuilt
    when the library is newer than its sources (the state gpurun ships to the GPU box)."""
    import subprocess
    csrc = os.path.join(PKG, "csrc")
    lib = os.path.join(PKG, "lpbox_hip", "liblpbox_hip.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "lpbox_hip.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", csrc, "all"])


def _gpu_available():
    try:
        from lpbox_hip import _lib
        return _lib.load().lpbox_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the HIP path is the product.
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
