"""Test-only stand-in for HipDistributedIndex._local_topk: the CPU oracle computes the shard-local top-k so
that the HOST logic (query gather, packing, cross-rank merge, passage exchange, persistence) can be exercised
without a GPU (gloo, CPU tensors). Never used by the product."""
import numpy as np
import torch

from oracle import oracle


def oracle_local_topk(self, q, k, pack=None):
    self._last_packed = None                  # (the host packing path is what these tests exercise)
    slab = self._slab.cpu().numpy()
    q16 = q.detach().cpu().to(torch.float16).numpy()
    s, i = oracle.search(q16, slab, k)
    self.last_search_stats = {"path": "oracle(test)"}
    return torch.from_numpy(s.copy()), torch.from_numpy(i.copy()), s, i
