"""GEMM-shaped passes of ONE column tile (<= 256 queries): slab DMA with the default cache policy against `nt` (gscan_kernel.h: NT; round 6), same process
(tuning build: atlas_tune_set_gscan_nt), alternated round by round, results compared bit for bit.
    python tools/gscan_nt_ab.py [rows ...] [--batches 96,128,192,256,384] [--certify]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import ctypes, sys, time
import numpy as np
import torch
from atlas_amd import _lib
from scan_policy_common import shard

L.atlas_tune_set_gscan_nt.argtypes, L.atlas_tune_set_gscan_nt.restype = [ctypes.c_int], None
D, k = 768, 40
args = sys.argv[1:]
flags = _lib.SCAN_TRUST_PMAX
if "--certify" in args:
    args.remove("--certify"); flags = 0
batches = (96, 128, 192, 256, 384)
if "--batches" in args:
    i = args.index("--batches"); batches = tuple(int(x) for x in args[i + 1].split(",")); del args[i:i + 2]
sizes = [int(a) for a in args] or [4_000_000]
stream = torch.cuda.current_stream().cuda_stream
for N in sizes:
    slab = shard(N)
    for B in batches:
        q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda").half()
        out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
        out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
        ws = torch.zeros(L.atlas_scan_topk_workspace_bytes(N, B, D, k), dtype=torch.uint8, device="cuda")
        reps = max(3, int((200 if N <= 4_000_000 else 24) * 64 / B))
        res, outs = {0: [], 1: []}, {}
        for rnd in range(4):
            for mode in (0, 1):
                L.atlas_tune_set_gscan_nt(mode)
                def call():
                    assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F16, slab.data_ptr(), N, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), stream, None, None, flags) == 0
                for _ in range(2): call()
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(reps): call()
                torch.cuda.synchronize()
                res[mode].append((time.perf_counter() - t) / reps * 1e3)
                assert int(out_st[_lib.ST_FLAGS]) == 0
                cur = (out_s.clone(), out_i.clone())
                if mode in outs: assert torch.equal(cur[0], outs[mode][0]) and torch.equal(cur[1], outs[mode][1])
                outs[mode] = cur
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        a, b = float(np.median(res[0])), float(np.median(res[1]))
        plan = _lib.decode_plan(int(out_st[_lib.ST_PLAN]))
        print(f"{N:>9d} rows, {B:4d} queries ({'certifying' if flags == 0 else 'trusting'}; plan {plan}): default policy {a:7.4f} ms   nt {b:7.4f} ms   ({(b / a - 1) * 100:+.1f} %)   identical: {same}", flush=True)
    del slab
    torch.cuda.empty_cache()
L.atlas_tune_set_gscan_nt(1)
