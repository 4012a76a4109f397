"""Streamed refresh from the token store: batches of 512 PASSAGES vs batches of 65 536 TOKENS (atlas_amd.refresh.TOKEN_BUDGET), alternated
in one process; the slab must come out bit-identical.   python tools/streamed_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atlas_amd import HipDistributedIndex, retrievers, refresh as refresh_mod
from atlas_amd.token_store import TokenStore

nb, n_s = 512, 512 * 32
rs = np.random.default_rng(4321)
lens = rs.integers(64, 201, size=n_s)
off = np.zeros(n_s + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
store = TokenStore(torch.from_numpy(rs.integers(1000, 30522, size=int(off[-1])).astype(np.int32)), off, 200)
torch.manual_seed(99)
enc = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
idx = HipDistributedIndex(); idx._set_slab(torch.zeros((n_s, 768), dtype=torch.float16, device="cuda"))
rf = refresh_mod.IndexRefresher(idx, enc, max_batch=nb, max_len=200, depth=3)
res, slabs = {0: [], refresh_mod.TOKEN_BUDGET: []}, {}
for rnd in range(4):
    for tb in res:
        rf.run_store(store, nb, token_budget=tb); torch.cuda.synchronize()
        t = time.perf_counter(); rf.run_store(store, nb, token_budget=tb, repeat=2); torch.cuda.synchronize()
        res[tb].append((time.perf_counter() - t) / 2)
        slabs[tb] = idx._slab.clone()
print("slabs identical:", torch.equal(slabs[0], slabs[refresh_mod.TOKEN_BUDGET]))
for tb, t in res.items():
    groups = store.plan(nb, True, tb)
    print(f"{'batches of %d tokens' % tb if tb else 'batches of %d passages' % nb:28s}: {len(groups):3d} batches, {np.median(t) * 1e3:8.2f} ms per refresh of {n_s} passages = {n_s / np.median(t):8.0f} passages/s", flush=True)
