import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# tests/test_gpu_multi.py stands eight device slots on the ONE GPU of the test box, and the ring form of the sharded sums
# lets their kernels wait for one another on the device: every slot needs a hardware queue of its own there (HIP's
# default is four per device, shared round robin by all streams).  Read by the HIP runtime when it starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden
