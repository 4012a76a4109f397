"""Where the reference checkout is present (the build container: /root/reference), run its OWN flat index
(src/index.py:113-120, imported unmodified through the shim of tests/golden/make_golden.py) on fresh random inputs --
other sizes, k, score magnitudes and forced ties than the committed fixtures -- and hold the oracle to it with the same
tie- / 1-ulp-aware comparison. Skipped on the GPU box, which has no /root/reference (the committed fixtures cover it there)."""
import os
import sys

import numpy as np
import pytest
import torch

import parity

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "index.py")), reason="reference checkout not present")

CASES = [
    # N, B, k, scale of the rows (fp16 exponent bucket of the scores), dup (forced ties), seed
    (1500, 3, 1, 1.0, 1, 101),
    (4097, 9, 40, 1.0, 1, 102),            # N not a multiple of any tile
    (2000, 5, 80, 0.05, 1, 103),           # over-retrieve k, small scores
    (900, 2, 128, 8.0, 1, 104),            # rerank k, large scores
    (1024, 4, 16, 1.0, 8, 105),            # every row 8 times: ties everywhere
    (60, 6, 60, 1.0, 1, 106),              # k == N
]


@pytest.fixture(scope="module")
def reference_index_cls():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden

    return make_golden.import_reference_index()


@pytest.mark.parametrize("N,B,k,scale,dup,seed", CASES)
def test_oracle_agrees_with_the_reference_run_live(N, B, k, scale, dup, seed, reference_index_cls, oracle_mod):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((N // dup, 768)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    P = np.tile((base * scale).astype(np.float16), (dup, 1))
    Q = rng.standard_normal((B, 768)).astype(np.float32)
    idx = reference_index_cls()
    idx.is_in_gpu = False
    idx.init_embeddings([{"id": str(i)} for i in range(P.shape[0])])
    idx.embeddings[:, :] = torch.from_numpy(P).T
    threads = torch.get_num_threads()
    torch.set_num_threads(1)        # threaded, the reference's CPU matmul is not run-to-run deterministic: both calls must see the same scores
    try:
        rs, ri = idx._compute_scores_and_indices(torch.from_numpy(Q), k)           # the reference's two torch calls
        xs, xi = idx._compute_scores_and_indices(torch.from_numpy(Q), min(P.shape[0], k + 24))   # ... and its scores around the cut
    finally:
        torch.set_num_threads(threads)
    s, i, full = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k, return_full=True)
    st = parity.compare_with_reference(rs.numpy().astype(np.float16), ri.numpy().astype(np.int64), xs.numpy().astype(np.float16),
                                       xi.numpy().astype(np.int64), full, s, i)
    print((N, B, k, scale, dup), st)
    assert st["max_ulp"] <= 1
    if dup == 1 and k < N:
        assert st["clean_rows"] >= 1, "the sharp rule (clean rows must be identical) was never exercised"
