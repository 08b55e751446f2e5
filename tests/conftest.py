import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# HIP streams share GPU_MAX_HW_QUEUES hardware queues (default 4) and streams on one queue serialise: tests that need two
# streams to run side by side (tests/test_gpu_fps_abort.py: a kernel holding the CUs beside the launch under test) would
# measure the queue, not the kernels.  Read by the runtime when it initialises, so set before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "tail: the test chooses the decoder's launch route itself (tests/test_gpu_decoder.py)")


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
