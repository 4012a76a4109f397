"""Build container only (needs /root/reference): times the REFERENCE CLASS ITSELF -- `DistributedIndex._compute_scores_and_indices`,
src/index.py:113-120, imported unmodified through the shim of tests/golden/make_golden.py -- on this host's cores, beside the port bench.py
times on the GPU box (oracle/ref_port.py: the same two torch calls), on the same rows, and checks that they return the same bits.
    python tools/ref_class_cpu_time.py [rows, default 1000000] > profiles/r04/cpu_reference_class_vs_port.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden  # noqa: E402
from oracle import ref_port  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B, k = 64, 40
cores = os.cpu_count()
torch.set_num_threads(cores)
g = torch.Generator().manual_seed(1234)
P = torch.empty((N, 768), dtype=torch.float16)
for r0 in range(0, N, 100_000):
    x = torch.randn((min(100_000, N - r0), 768), generator=g)
    P[r0:r0 + x.shape[0]] = (x / x.norm(dim=1, keepdim=True)).half()
Q = torch.randn((B, 768), generator=g)
cls = make_golden.import_reference_index()
idx = cls()
idx.is_in_gpu = False
idx.init_embeddings([{"id": str(i)} for i in range(N)])
idx.embeddings[:, :] = P.T


def med(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)), out


t_ref, (s_ref, i_ref) = med(lambda: idx._compute_scores_and_indices(Q, k))
emb = P.T.contiguous()
t_port, (s_port, i_port) = med(lambda: ref_port.reference_flat_search(Q, emb, k))
same = torch.equal(s_ref, s_port) and torch.equal(i_ref, i_port)
print(f"host: {cores} cores, torch {torch.__version__}, {N} rows x 768 fp16, {B} queries, top-{k}")
print(f"reference class  DistributedIndex._compute_scores_and_indices (src/index.py:113-120): {t_ref:8.3f} s per batch = {B / t_ref:8.2f} queries/s")
print(f"port             oracle/ref_port.reference_flat_search (what bench.py times)        : {t_port:8.3f} s per batch = {B / t_port:8.2f} queries/s")
print(f"ratio reference / port: {t_ref / t_port:.3f}; identical scores and ids: {same}")
