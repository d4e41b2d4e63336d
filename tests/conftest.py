import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# PyTorch (device buffers for the *_device entry points, torch.distributed) is imported before anything
# else touches the process: importing it AFTER the oracle's OpenMP pool and the HIP library are live has
# been seen to dead-lock inside torch/__init__.py on the GPU box (a test that imported it late hung).
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One GpuContext for the whole GPU session.  Fails loudly (no skip, no fallback) when the
    HIP library or the device is missing."""
    import threecrate_amd as tc
    c = tc.GpuContext(0)
    yield c
    c.close()
