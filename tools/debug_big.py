"""Find where the scan departs from the exact path as the shard grows (32M failed the bench's sanity check)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex

def shard(rows, seed=1234):
    g = torch.Generator(device="cuda").manual_seed(seed)
    slab = torch.empty((rows, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, rows, 250_000):
        n = min(250_000, rows - r0)
        x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0+n] = (x / x.norm(dim=1, keepdim=True)).half()
    return slab

q = torch.randn((64, 768), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
for N in [int(a) for a in sys.argv[1:]] or [8_000_000, 16_000_000, 32_000_000]:
    slab = shard(N)
    idx = HipDistributedIndex(); idx._set_slab(slab)
    s, i = idx._compute_scores_and_indices(q, 40)
    st = dict(idx.last_search_stats)
    es, ei = idx._exact_topk(q[:4], 40)
    same_i = torch.equal(i[:4], ei); same_s = torch.equal(s[:4], es)
    # direct fp64 check of the scan's own (row, score) pairs, row by row (no fancy indexing)
    rows = i[0].tolist()
    direct = torch.stack([slab[r].double() @ q[0].half().double() for r in rows]).half()
    gather = (slab[i[0]].double() @ q[0].half().double()).half()
    print(f"N={N}: scan==exact ids {same_i} scores {same_s}; scan scores==direct fp64 {torch.equal(direct, s[0])}; "
          f"fancy-index gather==direct {torch.equal(gather, direct)}; stats {st}", flush=True)
    if not same_i:
        bad = (i[:4] != ei).nonzero()
        print("  first mismatches:", bad[:5].tolist(), i[:4][i[:4] != ei][:5].tolist(), ei[i[:4] != ei][:5].tolist())
        print("  max row in scan result:", int(i.max()), "min", int(i.min()))
    del slab, idx
    torch.cuda.empty_cache()
