import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    # numpy's safe loader: no pickled objects inside
    with np.load(os.path.join(GOLDEN_DIR, "golden.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def ipx():
    from mural_amd import _lib
    return _lib.lib()


def gaussian(seed, shape):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def diag_build():
    """True when the loaded library is the diagnostic build (-DIPX_DIAG): it also carries the large-batch bf16 kernels
    and variants the default library does not ship, and the "bf16_*" knobs that select them."""
    from mural_amd import _lib
    return b"diag" in _lib.lib().ipx_version()
