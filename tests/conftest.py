import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with `-m gpu`; when no GPU is visible
    # they are skipped rather than failed so `pytest tests/` works on CPU.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """After every GPU test: drop what the test left in torch's caching allocator.  The
    engines size their buffers from the device's FREE memory (and `BruteForce._fit` decides
    from it whether the three-engine numpy-stream pipeline fits), so a test must not inherit
    the cached blocks of the full-size tests that ran before it."""
    yield
    if "gpu" not in request.keywords:
        return
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:
        pass
