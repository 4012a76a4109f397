import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure); built on demand."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_index_cls():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from atlas_amd import HipDistributedIndex

    return HipDistributedIndex
