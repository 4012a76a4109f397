"""A/B of the run-time tile pool at the end of the slab (tuning build): share of a workgroup's tiles that is not pre-assigned
(per mille) and its cap, 0 = static split. Kernel time (hipEvents around the scan) and whole-search time, settings alternated.

    python tools/scan_pool_ab.py 1000000 4000000 [32000000]
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys, time
import numpy as np
import torch
from atlas_amd import HipDistributedIndex, _lib

B, k, D = 64, 40, 768
SETTINGS = [(0, 16), (60, 16), (60, 32), (30, 48)]
for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000]:
    reps = 40 if N <= 4_000_000 else 12
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((N, D), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 250_000):
        n = min(250_000, N - r0); x = torch.randn((n, D), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
    L.atlas_tune_set_scan_pool(0, 16)
    idx = HipDistributedIndex(); idx._set_slab(slab)
    s0, i0 = idx._compute_scores_and_indices(q, k)
    ws, pmax = idx._ws, float(idx._pmax)
    out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs: a.record(); b.record()
    torch.cuda.synchronize()
    for rnd in range(3):
        for permille, cap in SETTINGS:
            L.atlas_tune_set_scan_pool(permille, cap)
            def call(ev=None):
                rc = L.atlas_scan_topk_ex(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(),
                                          out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream, ev[0].cuda_event if ev else None, ev[1].cuda_event if ev else None)
                assert rc == 0, rc
            for _ in range(5): call()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for it in range(reps): call(evs[it])
            torch.cuda.synchronize(); step = (time.perf_counter() - t0) / reps * 1e3
            st = out_st.cpu().numpy()
            ok = torch.equal(out_s, s0) and torch.equal(out_i, i0) and int(st[0]) == 0
            t = np.array([a.elapsed_time(b) for a, b in evs])
            print(f"N={N:9d} pool={permille:3d}/1000 cap {cap:3d}: scan mean {t.mean():.4f} min {t.min():.4f} ms ({N * 1536 / t.mean() / 1e9 / 8:.3f} of peak)   step {step:.4f} ms "
                  f"({N * 1536 / step / 1e9 / 8:.3f})  candidates {int(st[3])}  identical={ok}", flush=True)
    L.atlas_tune_set_scan_pool(60, 32)
    del slab, idx, ws; torch.cuda.empty_cache()
