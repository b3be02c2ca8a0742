import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_library():
    """Builds (if stale) and loads the HIP library; used by GPU and ABI tests."""
    from adacharge_amd.build import build_hip_library
    from adacharge_amd import backend

    build_hip_library(verbose=False)
    return backend.load_library()
