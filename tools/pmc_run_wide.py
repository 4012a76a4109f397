"""Workload for PMC passes over the scan's two instantiations: 64 queries (one 64-query pass) and 96 queries (one 96-query pass) on N rows."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
for r0 in range(0, N, 250_000):
    n = min(250_000, N - r0); x = torch.randn((n, 768), generator=g, device="cuda")
    slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
idx = HipDistributedIndex(); idx._set_slab(slab)
for B in (64, 96):
    q = torch.randn((B, 768), device="cuda")
    for _ in range(4):
        idx._compute_scores_and_indices(q, 40)
torch.cuda.synchronize()
print("done", idx.last_search_stats)
