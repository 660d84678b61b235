import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def seeded_sd():
    from oracle import unet_oracle as O
    return O.make_seeded_state_dict(1234)


@pytest.fixture(scope="session")
def hip_lib_built():
    """Make sure the in-tree HIP extension exists (hipcc cross-compiles on CPU-only hosts)."""
    from ai_based_frame_interpolation_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    return _native.LIB_PATH
