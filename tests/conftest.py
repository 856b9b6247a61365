import os
import subprocess
import sys

import pytest

# PyTorch-ROCm bundles its own HIP / HSA runtime.  Whichever runtime opens the GPU first in a process wins and the other
# then finds no device, so torch (used by a few GPU tests and by bench.py) is imported before libvolviz_hip.so is loaded:
# the library then binds to the runtime torch has already mapped (same SONAME) whichever test files are selected.
try:
    import torch  # noqa: F401
except Exception:                                                   # the CPU-only checks do not need it
    torch = None

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "volume-viz_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle and the product library exist (both are plain `make` targets;
    on the GPU box the prebuilt .so files travel with the snapshot)."""
    if not os.path.exists(os.path.join(REPO, "oracle", "_build", "libvvoracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(REPO, "volume-viz_amd", "lib", "libvolviz_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "volume-viz_amd")], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(REPO, "oracle", "_ref", "libvvref.so")):
        subprocess.call(["make", "-C", os.path.join(REPO, "oracle"), "ref"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="session")
def ctx():
    """A product context on cuda:0 (gpu tests only)."""
    import volviz_amd as vv
    c = vv.Context(0)
    yield c
    c.close()

