import os
import sys

import pytest

# the oracle's OpenMP regions are tiny in the tests: never oversubscribe a shared box
os.environ.setdefault("OMP_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_everything():
    """Build the HIP library (cross-compiles without a GPU) and the oracle before any test."""
    import admm_elastic_amd as pkg
    pkg.build_library()
    from oracle import oracle
    oracle.build()
    yield
