import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The C oracle is a few hundred lines of gcc: build it on demand.  The HIP
    # library is built by __graft_entry__.build() (hipcc) and only checked here.
    from oracle import cpu
    cpu.build()
    lib = os.path.join(ROOT, "slam_amd", "lib", "libslam_amd.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "slam_amd", "csrc")])


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
