"""A shard sized for the 288 GB part (round 6): N rows (default 128M = 196.6 GB of slab) on ONE GPU through the product call, all 64 queries against the
MFMA-free exact path, hipEvents around the scan kernel.    python tools/scan_big.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from atlas_amd import HipDistributedIndex, _lib
from scan_policy_common import shard

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128_000_000
B, k, D = 64, 40, 768
t0 = time.time()
slab = shard(N)
torch.cuda.synchronize()
print(f"{N} rows x 768 fp16 = {N * 1536 / 1e9:.1f} GB built in {time.time() - t0:.1f} s; free / total HBM {torch.cuda.mem_get_info()[0] / 1e9:.1f} / {torch.cuda.mem_get_info()[1] / 1e9:.1f} GB", flush=True)
q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
idx = HipDistributedIndex(); idx._set_slab(slab)
s, i = idx._compute_scores_and_indices(q, k)
st = dict(idx.last_search_stats)
print("search:", st, flush=True)
t1 = time.time()
es, ei = idx._exact_topk(q, k)
torch.cuda.synchronize()
print(f"exact path: {time.time() - t1:.2f} s; ids equal {torch.equal(i, ei)}, scores equal {torch.equal(s, es)}; largest returned row {int(i.max())}", flush=True)
L = _lib.lib()
ws, pmax = idx._ws, float(idx._pmax)
out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in evs:
    a.record(); b.record()
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
for flags, name in ((_lib.SCAN_TRUST_PMAX, "trusting"), (0, "certifying")):
    for it in range(12):
        ev = evs[it - 2] if it >= 2 else (None, None)
        assert L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream,
                                       ev[0].cuda_event if ev[0] else None, ev[1].cuda_event if ev[1] else None, flags) == 0
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    print(f"{name:10s} scan kernel mean {t.mean():.3f} ms (min {t.min():.3f}) = {N * 1536 / t.mean() / 1e9 / 8:.3f} of 8 TB/s; identical to the exact path: {torch.equal(out_s, es) and torch.equal(out_i, ei)}", flush=True)
