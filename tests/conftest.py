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


@pytest.fixture(autouse=True)
def _no_reference_stubs_survive_a_test():
    """VERDICT r04 weak #1c: a test that imports the reference through in-memory stubs (`faiss`, a fake `src.retrievers`) must take them out of
    sys.modules again -- otherwise the suite is green only in alphabetical order. Checked after every test (set up first, so torn down after the
    test's own monkeypatch)."""
    yield
    left = [m for m in sys.modules if m in ("src", "faiss") or m.startswith("src.") or m.startswith("faiss.")]
    assert not left, f"reference modules / stubs left in sys.modules: {left}"
