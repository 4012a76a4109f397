"""Host-side cost of the synchronous product calls (what atlas.py sees per batch): `_compute_scores_and_indices` and `search_knn`
against the device time of the same search, with a cProfile of the python side.

    python tools/host_overhead.py 1000000 [4000000]
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import cProfile, pstats, sys, time
import torch
from atlas_amd import HipDistributedIndex

B, k, D = 64, 40, 768
for N in [int(a) for a in sys.argv[1:]] or [1_000_000]:
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((N, D), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 250_000):
        n = min(250_000, N - r0); x = torch.randn((n, D), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, D), generator=g, device="cuda")
    idx = HipDistributedIndex(); idx._set_slab(slab)

    idx.doc_map = {i: {"id": i} for i in range(N)}        # a real dict, as the reference builds it (index.py:47)
    for _ in range(5): idx.search_knn(q, k)
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps): idx._compute_scores_and_indices(q, k)
    t_csi = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps): idx.search_knn(q, k)
    t_knn = (time.perf_counter() - t0) / reps * 1e3
    # device time of the same search, no host sync per call
    from atlas_amd import _lib
    L = _lib.lib()
    ws, pmax = idx._ws, float(idx._pmax)
    out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        L.atlas_scan_topk(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / reps * 1e3
    print(f"N={N}: device pipeline {t_dev:.4f} ms/search | _compute_scores_and_indices {t_csi:.4f} ms (+{t_csi - t_dev:.4f}) | search_knn {t_knn:.4f} ms (+{t_knn - t_dev:.4f})", flush=True)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(reps): idx.search_knn(q, k)
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
    del slab, idx; torch.cuda.empty_cache()
