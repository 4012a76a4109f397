"""Randomised cross-check of the GEMM-shaped passes (product build): random shard sizes, batch sizes, k, score scales, both twins, against the
MFMA-free exact path on the device (an independent implementation: tests/test_gpu_search.py::test_1m_scan_equals_exact_path).
    python tools/gscan_fuzz.py [cases, default 40] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atlas_amd.index import HipDistributedIndex

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260925)
g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
bad = 0
for c in range(cases):
    N = int(rng.choice([65536, 65536 + int(rng.integers(1, 256)), int(rng.integers(66000, 400000)), int(rng.integers(400000, 1500000))]))
    B = int(rng.choice([int(rng.integers(97, 129)), int(rng.integers(129, 193)), int(rng.integers(193, 257)), int(rng.integers(257, 385)), int(rng.integers(385, 513)),
                        int(rng.integers(513, 1025)), int(rng.integers(1025, 1400))]))
    k = int(rng.choice([1, 5, 40, 40, 40, 100, 256]))
    scale = float(rng.choice([1.0, 1.0, 0.05, 4.0]))
    slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 200_000):
        n = min(200_000, N - r0)
        x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True) * scale).half()
    if rng.random() < 0.3:                      # duplicated rows: ties across tiles
        slab[N // 2: N // 2 + min(1000, N // 4)] = slab[: min(1000, N // 4)]
    q = torch.randn((B, 768), generator=g, device="cuda") * float(rng.choice([1.0, 0.3, 3.0]))
    idx = HipDistributedIndex()
    idx.init_embeddings([None] * 0)
    idx._set_slab(slab)
    idx.doc_map = {}
    certify = bool(rng.random() < 0.4)
    idx.certify_every = 1 if certify else 64
    s, i = idx._compute_scores_and_indices(q, k)
    st = dict(idx.last_search_stats)
    es, ei = idx._exact_topk(q, k)
    ok = torch.equal(s, es) and torch.equal(i, ei)
    bad += not ok
    print(f"case {c:3d}: N {N:8d} B {B:5d} k {k:3d} scale {scale:4.2f} twin {'certifying' if not st.get('pmax_trusted') else 'trusting'} plan {st['plan']} fallback {st['fallback_queries']:3d} path {st['path']}: {'identical' if ok else 'MISMATCH'}", flush=True)
    del idx, slab
print("mismatches:", bad)
sys.exit(1 if bad else 0)
