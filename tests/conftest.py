import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """The HIP product path.  Fails loudly (no skip, no fallback) if the
    extension is not built or no GPU is visible under -m gpu."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from rfdnet_amd import _lib
    _lib.lib()
    return _lib
