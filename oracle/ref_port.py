"""CPU baseline: the reference's flat search arithmetic on host cores. TEST/BENCH INFRASTRUCTURE ONLY.

Restates src/index.py:113-120 with the same two torch calls on the same (768, n) fp16 layout:

    scores = torch.matmul(allqueries.half(), self.embeddings)        index.py:117
    scores, indices = torch.topk(scores, topk, dim=1)                index.py:118

(`/root/reference` does not exist on the GPU box, so the class itself cannot be imported there; in the build
container tests/golden/make_golden.py runs the real class through an import shim and the results agree.)
The reference has no CPU FAISS path: this torch path IS its flat index.
"""
import os
import time

import torch


def reference_flat_search(q: torch.Tensor, embeddings_dN: torch.Tensor, topk: int):
    scores = torch.matmul(q.half(), embeddings_dN)
    return torch.topk(scores, topk, dim=1)


def time_reference_flat(slab_rows: torch.Tensor, q: torch.Tensor, topk: int, budget_s: float) -> dict:
    """slab_rows: (n, 768) fp16 CPU sample of the workload; returns the cpu_baseline JSON object."""
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    emb = slab_rows.T.contiguous()          # the reference's (d, n) layout (index.py:51)
    n = emb.shape[1]
    reference_flat_search(q, emb, topk)     # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 50):
        t0 = time.perf_counter()
        reference_flat_search(q, emb, topk)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": q.shape[0] / med, "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": f"first {n} rows of the workload x 768 fp16, {q.shape[0]} queries, top-{topk}; "
                  f"torch.matmul(fp16)+torch.topk (src/index.py:117-118), median of {len(times)} runs, "
                  f"{torch.get_num_threads()} threads",
        "seconds_per_batch": med, "rows": n,
    }
