"""Does recording the two hipEvents around the scan kernel (atlas_scan_topk_ex) change the step time? Same loop with and without them.

    python tools/event_overhead.py 1000000 4000000
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time
import torch
from atlas_amd import HipDistributedIndex, _lib

B, k, D, reps = 64, 40, 768, 200
L = _lib.lib()
for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000]:
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((N, D), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 250_000):
        n = min(250_000, N - r0); x = torch.randn((n, D), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, D), generator=g, device="cuda")
    idx = HipDistributedIndex(); idx._set_slab(slab)
    idx._compute_scores_and_indices(q, k)
    ws, pmax = idx._ws, float(idx._pmax)
    out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs: a.record(); b.record()
    torch.cuda.synchronize()

    def loop(with_events):
        for _ in range(10):
            L.atlas_scan_topk_ex(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream, None, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(reps):
            ev = evs[it] if with_events else None
            L.atlas_scan_topk_ex(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream,
                                 ev[0].cuda_event if ev else None, ev[1].cuda_event if ev else None)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    for rnd in range(3):
        a, b = loop(False), loop(True)
        print(f"N={N}: step without events {a:.4f} ms, with events {b:.4f} ms (+{(b - a) * 1e3:.1f} us)", flush=True)
    del slab, idx; torch.cuda.empty_cache()
