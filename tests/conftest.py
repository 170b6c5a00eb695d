import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ops_golden():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


@pytest.fixture(scope="session")
def netg_golden():
    return np.load(os.path.join(GOLDEN, "netg.npz"))


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip():
    """The C-ABI library on a GPU box (gpu tests only).  Fails loudly if the HIP library is missing."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from pwstablenet_amd import hipabi
    return hipabi
