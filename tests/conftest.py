import gc
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "subspace-reg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _release_device_memory():
    """Drop what a finished test left on the device: backbones (6 GB of workspaces each, lane streams), session buffers,
    hipGraphs.  Idle device first, then the Python references, then torch's cache."""
    torch = sys.modules.get("torch")
    if torch is None or not torch.cuda.is_available():
        return
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(autouse=True)
def _gpu_test_teardown(request):
    yield
    if request.node.get_closest_marker("gpu") is not None:
        _release_device_memory()


@pytest.fixture(scope="session", autouse=True)
def _gpu_session_teardown():
    """End of the test session: leave the GPU idle and empty, so that whatever runs next in the lease (smoke(), bench.py)
    starts on a clean device and this process exits without live streams / graphs."""
    yield
    _release_device_memory()
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.synchronize()
