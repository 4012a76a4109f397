import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex, _lib
L = _lib.lib()
L.atlas_tune_set_merge_stamps.argtypes = [ctypes.c_void_p]
def shard(rows, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    slab = torch.empty((rows, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, rows, 250_000):
        n = min(250_000, rows - r0); x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0+n] = (x / x.norm(dim=1, keepdim=True)).half()
    return slab
for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000]:
    slab = shard(N); q = torch.randn((64, 768), device="cuda")
    idx = HipDistributedIndex(); idx._set_slab(slab)
    idx._compute_scores_and_indices(q, 40)
    dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
    L.atlas_tune_set_merge_stamps(dbg.data_ptr())
    idx._compute_scores_and_indices(q, 40); torch.cuda.synchronize()
    L.atlas_tune_set_merge_stamps(None)
    t = dbg.cpu().tolist()
    names = ["init+scan", "keyload", "bitsearch", "band", "rescore", "rank"]
    print(N, idx.last_search_stats["candidates"], {n: (t[i+1]-t[i]) for i, n in enumerate(names)}, "total cycles", t[6]-t[0], "(100 MHz ticks?)")
