"""What would a slab pass over 80 / 96 queries cost on the compute side? (tuning build, round 3)
The 64-query pass is HBM-bound with the matrix pipe 17 % busy; a 96-query pass would read the slab once per 96 queries (+50 % queries per
byte) IF the extra 2 MFMAs + 2 LDS reads per k-step fit under the stream -- the scan runs at the board's power limit. The proxies compute
1 / 2 extra 16-query fragments per k-step from other image rows and throw them away (scan_kernel<16,1,8,64,QX>): same bytes, the matrix-pipe,
LDS-read and register load of the bigger pass, none of its LDS footprint (a real 96-query image is 147 KiB: 12 KiB left for candidates).
    python tools/scan_q96_proxy.py [rows ...]
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys, time
import numpy as np
import torch
from atlas_amd import _lib

B, k, D = 64, 40, 768
sizes = [int(a) for a in sys.argv[1:]] or [4_000_000, 32_000_000]
NMAX = max(sizes)
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((NMAX, D), dtype=torch.float16, device="cuda")
for r0 in range(0, NMAX, 1_000_000):
    n = min(1_000_000, NMAX - r0); x = torch.randn((n, D), generator=g, device="cuda"); slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
variants = [(0, "64 queries (product, trusted pmax)", 64), (7, "proxy: 80 queries' MFMAs + LDS reads", 80), (8, "proxy: 96 queries' MFMAs + LDS reads", 96)]
for N in sizes:
    reps = 100 if N <= 4_000_000 else 20
    ws = torch.zeros(L.atlas_scan_topk_workspace_bytes(N, B, D, k), dtype=torch.uint8, device="cuda")
    res = {v[0]: [] for v in variants}
    for rnd in range(5):
        for vi, name, nq in variants:
            L.atlas_tune_set_scan_variant(vi)
            def call():
                rc = L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                             ws.data_ptr(), ws.numel(), stream, None, None, _lib.SCAN_TRUST_PMAX)
                assert rc == 0, rc
            for _ in range(5): call()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps): call()
            torch.cuda.synchronize()
            res[vi].append((time.perf_counter() - t) / reps * 1e3)
    base = float(np.median(res[0]))
    for vi, name, nq in variants:
        med = float(np.median(res[vi]))
        print(f"{N:>9d} rows  {name:42s}: {med:8.4f} ms per pass = {N * 1536 / (med * 1e-3) / 1e12:5.2f} TB/s;  queries per second of slab time vs the 64-query pass: x {nq / 64 * base / med:5.3f}", flush=True)
L.atlas_tune_set_scan_variant(0)
