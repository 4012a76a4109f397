"""Batches above 64 queries: 64-query slab passes only vs + 96-query passes vs + PAIRED passes (two chunks concurrently on half the chip each: the product), same process (tuning build).
    python tools/batch_ab.py [rows, default 4000000 and 32000000]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys, time
import numpy as np
import torch
from atlas_amd import _lib

k, D = 40, 768
sizes = [int(a) for a in sys.argv[1:]] or [4_000_000, 32_000_000]
NMAX = max(sizes)
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((NMAX, D), dtype=torch.float16, device="cuda")
for r0 in range(0, NMAX, 1_000_000):
    n = min(1_000_000, NMAX - r0); x = torch.randn((n, D), generator=g, device="cuda"); slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
stream = torch.cuda.current_stream().cuda_stream
for N in sizes:
    for B in (64, 96, 128, 160, 192, 256, 512):
        q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda").half()
        out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
        out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
        ws = torch.zeros(L.atlas_scan_topk_workspace_bytes(N, B, D, k), dtype=torch.uint8, device="cuda")
        reps = max(3, int((200 if N <= 4_000_000 else 24) * 64 / B))
        res, outs = {0: [], 1: [], 2: []}, {}
        for rnd in range(3):
            for wide in (0, 1, 2):
                L.atlas_tune_set_scan_wide(1 if wide else 0); L.atlas_tune_set_scan_pair(1 if wide == 2 else 0)
                def call():
                    rc = L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F16, slab.data_ptr(), N, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), stream, None, None, _lib.SCAN_TRUST_PMAX)
                    assert rc == 0, rc
                for _ in range(2): call()
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(reps): call()
                torch.cuda.synchronize()
                res[wide].append((time.perf_counter() - t) / reps * 1e3)
                assert int(out_st[_lib.ST_FLAGS]) == 0, out_st[:8].tolist()
                cur = (out_s.clone(), out_i.clone())
                if wide in outs: assert torch.equal(cur[0], outs[wide][0]) and torch.equal(cur[1], outs[wide][1])
                outs[wide] = cur
        same = all(torch.equal(outs[0][0], outs[m][0]) and torch.equal(outs[0][1], outs[m][1]) for m in (1, 2))
        a, b, c = (float(np.median(res[m])) for m in (0, 1, 2))
        print(f"{N:>9d} rows, {B:3d} queries: 64-query passes {a:8.3f} ms = {B / a * 1e3:8.0f} queries/s;  + 96-query passes {b:8.3f} ms = {B / b * 1e3:8.0f} (x {a / b:5.3f});  + paired passes {c:8.3f} ms = {B / c * 1e3:8.0f} queries/s (x {a / c:5.3f})  identical results: {same}", flush=True)
L.atlas_tune_set_scan_wide(1); L.atlas_tune_set_scan_pair(1)
