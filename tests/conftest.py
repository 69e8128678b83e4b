import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """Full-size GPU tests size themselves against the free HBM and skip when it is short: give every GPU test a clean
    slate (tensors kept alive by an earlier failure's traceback, cached kernel workspaces, the allocator's cache)."""
    if "gpu" in request.keywords:
        import gc

        import torch

        if torch.cuda.is_available():
            from vivit_amd import kernels

            gc.collect()
            kernels._WORKSPACES.clear()
            torch.cuda.empty_cache()
    yield
