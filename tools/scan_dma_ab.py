"""The 64-query pass on scan_kernel.h (round 5: HBM -> VGPR in the MFMA fragment shape) against dscan_kernel.h (round 6: LDS-DMA `nt`, queries in
registers), same process, alternated round by round (boxes differ by several per cent; tuning build: atlas_tune_set_scan_dma 0 | 1 | 2 = default cache
policy on the DMA). Per mode and twin (trusting / certifying): hipEvents around the scan kernel and the whole search step; results must be bit-identical.

    python tools/scan_dma_ab.py 1000000 4000000 32000000 [--pool 60,32 --pool 200,32,128 ... = permille, cap[, rows of a pool tile of the DMA kernel]] [--modes 0,1,2]
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex, _lib  # noqa: E402
from scan_policy_common import shard  # noqa: E402

L.atlas_tune_set_scan_dma.argtypes, L.atlas_tune_set_scan_dma.restype = [ctypes.c_int], None
L.atlas_tune_set_dma_pool_tile.argtypes, L.atlas_tune_set_dma_pool_tile.restype = [ctypes.c_int], None
L.atlas_tune_set_dma_deal.argtypes, L.atlas_tune_set_dma_deal.restype = [ctypes.c_int], None
argv = sys.argv[1:]
pools, modes, sizes = [], [0, 1], []
while argv:
    a = argv.pop(0)
    if a == "--pool":
        pools.append(tuple(int(x) for x in argv.pop(0).split(",")))
    elif a == "--modes":
        modes = [int(x) for x in argv.pop(0).split(",")]
    else:
        sizes.append(int(a))
sizes = sizes or [4_000_000]
pools = pools or [(60, 32)]
reps = int(os.environ.get("REPS", "20"))
rounds = int(os.environ.get("ROUNDS", "3"))
B, k, D = int(os.environ.get("QUERIES", "64")), int(os.environ.get("K", "40")), 768      # (K=128: the over-retrieval of retrieve_with_rerank)
names = {0: "scan_kernel<16,1,8>", 1: "dscan_kernel<nt>", 2: "dscan_kernel<default>", 3: "dscan<nt> contiguous"}      # 3: one contiguous range per workgroup instead of dealt tiles
for N in sizes:
    slab = shard(N)
    q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
    L.atlas_tune_set_scan_dma(0)
    L.atlas_tune_set_scan_pool(60, 32)
    idx = HipDistributedIndex()
    idx._set_slab(slab)
    s0, i0 = idx._compute_scores_and_indices(q, k)
    ws, pmax = idx._ws, float(idx._pmax)
    out_s = torch.empty((B, k), dtype=torch.float16, device="cuda")
    out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); b.record()
    torch.cuda.synchronize()
    acc = {}
    for rnd in range(rounds):
        for pool in pools:
            L.atlas_tune_set_scan_pool(*pool[:2])
            L.atlas_tune_set_dma_pool_tile(pool[2] if len(pool) > 2 else 256)
            for flags in (_lib.SCAN_TRUST_PMAX, 0):
                for m in modes:
                    L.atlas_tune_set_scan_dma(1 if m == 3 else m)
                    L.atlas_tune_set_dma_deal(0 if m == 3 else 1)
                    for it in range(3):
                        assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), stream, None, None, flags) == 0
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for it in range(reps):
                        assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), stream, evs[it][0].cuda_event, evs[it][1].cuda_event, flags) == 0
                    torch.cuda.synchronize()
                    step = (time.perf_counter() - t0) / reps * 1e3
                    ok = torch.equal(out_s, s0) and torch.equal(out_i, i0) and int(out_st[_lib.ST_FLAGS]) == 0
                    t = np.array([a.elapsed_time(b) for a, b in evs])
                    key = (pool, flags, m)
                    acc.setdefault(key, []).append((t.mean(), t.min(), step, ok, int(out_st[_lib.ST_N_CANDIDATES])))
    for key, v in acc.items():
        pool, flags, m = key
        a = np.array([(x[0], x[1], x[2]) for x in v])
        pool_s = ",".join(str(x) for x in pool)
        print(f"N={N:9d} pool={pool_s:>10s} {'trusting  ' if flags else 'certifying'} {names[m]:22s} kernel mean {a[:, 0].mean():.4f} (min {a[:, 1].min():.4f}) ms = "
              f"{N * 1536 / a[:, 0].mean() / 1e9 / 8:.3f} of 8 TB/s   step {a[:, 2].mean():.4f} ms = {N * 1536 / a[:, 2].mean() / 1e9 / 8:.3f}   candidates {v[-1][4]}   "
              f"identical={all(x[3] for x in v)}", flush=True)
    L.atlas_tune_set_dma_pool_tile(256); L.atlas_tune_set_dma_deal(1)
    del slab, idx, ws
    torch.cuda.empty_cache()
