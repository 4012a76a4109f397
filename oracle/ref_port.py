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


def _time(fn, min_runs, budget_s, max_runs=20):
    fn()                                     # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < min_runs or (time.perf_counter() < t_end and len(times) < max_runs):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def time_reference_flat(slab_rows: torch.Tensor, q: torch.Tensor, topk: int, budget_s: float, workload_rows: int = None,
                        min_rows: int = 500_000) -> dict:
    """slab_rows: (n, 768) fp16 CPU rows of the workload; returns the cpu_baseline JSON object.

    The sample is at least `min_rows` rows (less only if the workload is smaller) and otherwise bounded to about `budget_s` of CPU
    work: a 20k-row probe sets the sample size. torch's CPU half matmul is slow (~18 us per row and 64 queries on a 256-core host), so
    the big sample is timed ONCE after a warm-up on the probe (it is seconds long: run-to-run noise is far below the GPU / CPU ratio
    it is there to put a scale on). `value` is queries/s scaled to `workload_rows` (time is linear in rows: one GEMM column and one
    top-k element per row); the raw measurement is kept next to it.
    """
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    n_all = slab_rows.shape[0]
    workload_rows = workload_rows or n_all
    probe_n = min(20_000, n_all)
    probe = slab_rows[:probe_n].T.contiguous()          # the reference's (d, n) layout (index.py:51)
    t_probe, _ = _time(lambda: reference_flat_search(q, probe, topk), 1, 0.0)
    n = int(min(n_all, max(min_rows, probe_n * (budget_s * 0.6) / max(t_probe, 1e-6))))
    emb = slab_rows[:n].T.contiguous()
    t0 = time.perf_counter()
    reference_flat_search(q, emb, topk)
    med, runs = time.perf_counter() - t0, 1
    if med < budget_s * 0.25:                            # a fast host: take the median of a few more
        med, runs = _time(lambda: reference_flat_search(q, emb, topk), 2, budget_s * 0.4)
    # secondary line: the same two calls in fp32 ("FAISS-equivalent arithmetic", NOT FAISS; BASELINE.md §3 line B)
    emb32 = emb.float()
    q32 = q.float()
    med32, runs32 = _time(lambda: torch.topk(torch.matmul(q32, emb32), topk, dim=1), 1, budget_s * 0.1)
    return {
        "value": q.shape[0] / med * (n / workload_rows), "unit": "queries/s", "cores": cores,
        "kind": "port" if n >= workload_rows else f"port, extrapolated from {n} rows",
        "sample": f"first {n} rows of the workload x 768 fp16, {q.shape[0]} queries, top-{topk}; "
                  f"torch.matmul(fp16)+torch.topk (src/index.py:117-118) on the host, {'one run' if runs == 1 else 'median of %d runs' % runs} after a "
                  f"warm-up on {probe_n} rows, {torch.get_num_threads()} threads; value = measured rate x {n}/{workload_rows} rows",
        "measured_queries_per_s_on_sample": q.shape[0] / med, "seconds_per_batch_on_sample": med, "rows": n,
        "fp32_arith_queries_per_s_scaled": q.shape[0] / med32 * (n / workload_rows),
        "fp32_note": "same two torch calls in fp32 (what FAISS IndexFlatIP computes); NOT FAISS (not installed, not a reference path)",
    }
