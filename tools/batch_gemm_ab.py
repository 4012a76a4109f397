"""Batches above 96 queries: the streaming passes of round 3 (64 / 96-query passes, single or paired) vs the GEMM-shaped passes (gscan_kernel.h), same
process (tuning build), results compared bit for bit.
    python tools/batch_gemm_ab.py [rows, default 4000000 and 32000000] [--k K] [--batches 97,128,...] [--modes 0,1] [--certify]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys, time
import numpy as np
import torch
from atlas_amd import _lib

D = 768
args = sys.argv[1:]
k = 40
if "--k" in args:
    i = args.index("--k"); k = int(args[i + 1]); del args[i:i + 2]
modes = (0, 1)
if "--modes" in args:      # 0 = streaming passes, 1 = gscan_kernel (one mode: timing only, e.g. for library variants via ATLAS_HIP_SO)
    i = args.index("--modes"); modes = tuple(int(x) for x in args[i + 1].split(",")); del args[i:i + 2]
flags = _lib.SCAN_TRUST_PMAX
if "--certify" in args:    # the C-ABI's default contract: every row norm measured beside the MFMAs (gscan_kernel<2, .>)
    args.remove("--certify"); flags = 0
batches = (64, 96, 128, 192, 256, 384, 512, 1024)
if "--batches" in args:
    i = args.index("--batches"); batches = tuple(int(x) for x in args[i + 1].split(",")); del args[i:i + 2]
sizes = [int(a) for a in args] or [4_000_000, 32_000_000]
NMAX = max(sizes)
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((NMAX, D), dtype=torch.float16, device="cuda")
for r0 in range(0, NMAX, 1_000_000):
    n = min(1_000_000, NMAX - r0); x = torch.randn((n, D), generator=g, device="cuda"); slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
stream = torch.cuda.current_stream().cuda_stream
for N in sizes:
    for B in batches:
        q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda").half()
        out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
        out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
        ws = torch.zeros(L.atlas_scan_topk_workspace_bytes(N, B, D, k), dtype=torch.uint8, device="cuda")
        reps = max(3, int((200 if N <= 4_000_000 else 24) * 64 / B))
        res, outs, stats = {m: [] for m in modes}, {}, {}
        for rnd in range(3):
            for mode in modes:
                L.atlas_tune_set_scan_gemm(mode)
                def call():
                    rc = L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F16, slab.data_ptr(), N, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), stream, None, None, flags)
                    assert rc == 0, rc
                for _ in range(2): call()
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(reps): call()
                torch.cuda.synchronize()
                res[mode].append((time.perf_counter() - t) / reps * 1e3)
                st = out_st[:8].tolist()
                assert st[_lib.ST_FLAGS] == 0, st
                stats[mode] = (st[_lib.ST_N_CANDIDATES] / B, st[_lib.ST_N_RESCORED] / B, float(np.array(st[_lib.ST_MAXERR_BITS], dtype=np.int32).view(np.float32)))
                cur = (out_s.clone(), out_i.clone())
                if mode in outs: assert torch.equal(cur[0], outs[mode][0]) and torch.equal(cur[1], outs[mode][1])
                outs[mode] = cur
        same = all(torch.equal(outs[modes[0]][0], outs[m][0]) and torch.equal(outs[modes[0]][1], outs[m][1]) for m in modes)
        if len(modes) != 2:
            print(f"{N:>9d} rows, {B:4d} queries, k {k}: " + "  ".join(f"mode {m}: {float(np.median(res[m])):8.3f} ms" for m in modes) + f"  identical: {same}", flush=True)
            continue
        a, b = (float(np.median(res[m])) for m in modes)
        tf = 2.0 * B * N * D / (b * 1e-3) / 1e12
        print(f"{N:>9d} rows, {B:4d} queries, k {k}: streaming passes {a:8.3f} ms = {B / a * 1e3:8.0f} q/s;  GEMM-shaped {b:8.3f} ms = {B / b * 1e3:8.0f} q/s (x {a / b:5.3f}; {tf:6.0f} TFLOP/s = {tf / 2500:5.3f} of the f16 MFMA peak; "
              f"{N * 1536 / (b * 1e-3) / 1e12:5.2f} TB/s of slab per pass-set); candidates/query {stats[modes[0]][0]:.0f} vs {stats[modes[1]][0]:.0f}, rescored {stats[modes[0]][1]:.1f} vs {stats[modes[1]][1]:.1f}, max err/eps {stats[modes[1]][2]:.3f}; identical: {same}", flush=True)
L.atlas_tune_set_scan_gemm(1)
