"""A/B of the two modes of the scan in one process (product library): certifying (atlas_scan_topk_ex: every row's norm measured in the
scan, 4 v_dot2 per MFMA) vs trusting a certified pmax (atlas_scan_topk_flags + ATLAS_SCAN_TRUST_PMAX). Kernel time (hipEvents), step
time, identical results.

    python tools/scan_trust_ab.py 1000000 4000000 32000000
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time
import numpy as np
import torch
from atlas_amd import HipDistributedIndex, _lib

L = _lib.lib()
B, k, D = 64, 40, 768
for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000]:
    reps = 60 if N <= 4_000_000 else 20
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((N, D), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 250_000):
        n = min(250_000, N - r0); x = torch.randn((n, D), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
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
        for name, flags in (("certify", 0), ("trust  ", _lib.SCAN_TRUST_PMAX)):
            def call(ev=None):
                rc = L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                             ws.data_ptr(), ws.numel(), stream, ev[0].cuda_event if ev else None, ev[1].cuda_event if ev else None, flags)
                assert rc == 0, rc
            for _ in range(5): call()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for it in range(reps): call()
            torch.cuda.synchronize(); step = (time.perf_counter() - t0) / reps * 1e3
            for it in range(reps): call(evs[it])
            torch.cuda.synchronize()
            st = out_st.cpu().numpy()
            ok = torch.equal(out_s, s0) and torch.equal(out_i, i0) and int(st[0]) == 0
            t = np.array([a.elapsed_time(b) for a, b in evs])
            print(f"N={N:9d} {name}: scan mean {t.mean():.4f} ms ({N * 1536 / t.mean() / 1e9 / 8:.3f} of peak)   step {step:.4f} ms ({N * 1536 / step / 1e9 / 8:.3f})  identical={ok}", flush=True)
    del slab, idx, ws; torch.cuda.empty_cache()
