import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "data")

# One HIP runtime per test process.  torch ships its own libamdhip64 / librccl (same sonames as /opt/rocm's, other builds) and
# asks for them by a name the loader does not match against an already loaded /opt/rocm copy: a process that loads
# libdpgo_hip.so (hence /opt/rocm's runtime) FIRST and imports torch LATER ends up with two runtimes and aborts at exit
# ("double free or corruption": tests/test_gpu_rank_exchange.py in front of tests/test_gpu_distributed.py).  With torch
# imported first, libdpgo_hip.so binds to torch's copies through their sonames -- whatever order the tests run in.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def data_dir():
    return DATA
