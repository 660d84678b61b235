import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU tier under the driver's `-x`: the oracle / golden parity tests first, single-process
# feature tests next, multi-process rehearsals (timing- and transport-dependent) LAST, so that a failure in
# a rehearsal can cost at most the rehearsal tests.  Files not listed keep their place in the middle.
_FILE_ORDER = ["test_oracle", "test_abi", "test_host", "test_gpu_parity", "test_gpu_configs", "test_gpu_cabi",
               "test_gpu_tiling", "test_metrics", "test_gpu_serving"]
_LAST = ["test_dist", "test_gpu_dist"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FILE_ORDER:
            return _FILE_ORDER.index(name)
        if name in _LAST:
            return 1000 + _LAST.index(name)
        return 500
    items.sort(key=key)   # stable: the order inside a file is untouched


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
