"""per-workgroup cycle stamps of the scan kernel (dev tool): distribution of start-up, loop and end times over the 256 CUs"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex, _lib
import numpy as np
L = _lib.lib()
L.atlas_tune_set_scan_stamps.argtypes = [ctypes.c_void_p]
for N in [int(a) for a in sys.argv[1:]] or [4_000_000, 1_000_000]:
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 250_000):
        n = min(250_000, N - r0); x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0+n] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((64, 768), device="cuda")
    idx = HipDistributedIndex(); idx._set_slab(slab)
    for _ in range(3): idx._compute_scores_and_indices(q, 40)
    dbg = torch.zeros(256 * 4, dtype=torch.int64, device="cuda")
    res = []
    for rep in range(5):
        dbg.zero_()
        L.atlas_tune_set_scan_stamps(dbg.data_ptr())
        idx._compute_scores_and_indices(q, 40); torch.cuda.synchronize()
        L.atlas_tune_set_scan_stamps(None)
        t = dbg.cpu().numpy().reshape(256, 4).astype(np.float64)
        t0 = t[:, 0].min()
        res.append(t - t0)
    t = res[-1]
    f = lambda a: "min %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (a.min(), np.median(a), np.percentile(a, 90), a.max())
    print(f"N={N}  (s_memtime ticks; kernel span = {t[:,3].max():.0f})")
    print("  entry          ", f(t[:, 0]))
    print("  after LDS fill ", f(t[:, 1] - t[:, 0]), "(duration)")
    print("  loop           ", f(t[:, 2] - t[:, 1]), "(duration)")
    print("  hand-over      ", f(t[:, 3] - t[:, 2]), "(duration)")
    print("  end            ", f(t[:, 3]))
    order = np.argsort(t[:, 3])
    print("  slowest WGs", order[-8:].tolist(), "their loop durations", (t[order[-8:], 2] - t[order[-8:], 1]).astype(int).tolist())
    byxcd = [(t[x::8, 2] - t[x::8, 1]).mean() for x in range(8)]
    print("  mean loop duration per XCD (wg % 8):", [int(v) for v in byxcd])
    spans = [r[:, 3].max() for r in res]; print("  spans over 5 reps:", [int(v) for v in spans], " p50 end over reps:", [int(np.median(r[:, 3])) for r in res])
