import os
import sys

import pytest

# The CPU oracle (OpenMP, one thread per hardware thread of the GPU box) shares the test process with torch's own OpenMP pool:
# two pools of ~256 busy-waiting threads starve each other (a 64x64 oracle forward took 23 s instead of 0.3 s after the first
# oracle trace had started its pool).  Idle OpenMP threads sleep instead of spinning; set before either runtime is loaded.
# (Only where a GPU is present: on the 8-core build container the pools do not collide and sleeping threads make the OpenMP
# oracle five times slower.)
if os.path.exists("/dev/kfd"):
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
    os.environ.setdefault("KMP_BLOCKTIME", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
