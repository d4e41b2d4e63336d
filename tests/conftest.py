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
    # The libraries are build products (git-ignored; __graft_entry__.build() / `make -C threecrate_amd/csrc` make them, the GPU box gets them
    # with the snapshot).  A checkout that has not been built yet is built here, once, where the compiler is at hand -- the shipped library,
    # the development build the fault-injection / debug-bit tests load (variants/libthreecrate_hip_dev.so) and the oracle.
    import shutil
    import subprocess
    need = [os.path.join(ROOT, "threecrate_amd", "libthreecrate_hip.so"), os.path.join(ROOT, "threecrate_amd", "variants", "libthreecrate_hip_dev.so")]
    if not all(os.path.exists(p) for p in need) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "threecrate_amd", "csrc"), "-j4"], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libtc_oracle.so")) and shutil.which("gcc"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def ctx():
    """One GpuContext for the whole GPU session.  Fails loudly (no skip, no fallback) when the
    HIP library or the device is missing."""
    import threecrate_amd as tc
    c = tc.GpuContext(0)
    yield c
    c.close()
