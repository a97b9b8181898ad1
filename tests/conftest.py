import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _fork_every_block(request):
    """GPU tests run small models: let every block fork its weight gradients to the second stream (production skips
    the fork for blocks too small to repay it), so that the two-stream path is what the parity tests exercise."""
    if "gpu" not in request.keywords:
        yield
        return
    from mia_amd import ops
    old = ops.Side.MIN_WORK
    ops.Side.MIN_WORK = 0
    yield
    ops.Side.MIN_WORK = old
