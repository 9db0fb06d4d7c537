import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "data")

# One HIP runtime per test process: dpgo_ros_amd.capi.lib() imports torch (where it exists) in front of libdpgo_hip.so, so
# the library binds to torch's copies of libamdhip64 / librccl whatever order the tests run in (see capi.lib).


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def data_dir():
    return DATA
