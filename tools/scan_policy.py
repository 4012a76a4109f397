"""Scan-kernel variants in one process: the slab is built once, every variant (atlas_tune_set_scan_variant, tuning build) runs the
full C-ABI search `reps` times; hipEvents around the scan kernel (atlas_scan_topk_ex), results must be bit-identical.

    python tools/scan_policy.py 4000000 32000000 -- 0 5 6
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex, _lib  # noqa: E402


def shard(rows, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    slab = torch.empty((rows, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, rows, 250_000):
        n = min(250_000, rows - r0)
        x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    return slab


argv = sys.argv[1:]
if "--" in argv:
    i = argv.index("--")
    sizes, variants = [int(a) for a in argv[:i]], [int(a) for a in argv[i + 1:]]
else:
    sizes, variants = [int(a) for a in argv] or [4_000_000], [0, 5]
reps = int(os.environ.get("REPS", "20"))
rounds = int(os.environ.get("ROUNDS", "2"))
L = _lib.lib()
B, k, D = 64, 40, 768
for N in sizes:
    slab = shard(N)
    q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
    L.atlas_tune_set_scan_variant(0)
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
    for rnd in range(rounds):
        for v in variants:
            L.atlas_tune_set_scan_variant(int(v))
            name = L.atlas_build_info().decode().split()[2]
            for it in range(3 + reps):
                ev = evs[it - 3] if it >= 3 else None
                rc = L.atlas_scan_topk_ex(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(),
                                          out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream,
                                          ev[0].cuda_event if ev else None, ev[1].cuda_event if ev else None)
                assert rc == 0, rc
            torch.cuda.synchronize()
            ok = torch.equal(out_s, s0) and torch.equal(out_i, i0) and int(out_st[0]) == 0
            t = np.array([a.elapsed_time(b) for a, b in evs])
            print(f"N={N:9d} v={v:2d} {name:28s} scan mean {t.mean():.4f} min {t.min():.4f} max {t.max():.4f} ms  "
                  f"{N * 1536 / t.mean() / 1e9:.3f} TB/s  frac {N * 1536 / t.mean() / 1e9 / 8:.3f}  identical={ok}", flush=True)
    del slab, idx, ws
    torch.cuda.empty_cache()
