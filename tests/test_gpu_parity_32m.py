"""Parity at the size the headline number is quoted on: 32M passages x 768 fp16 on one GPU (BASELINE.json north_star target;
49.2 GB of the 288 GB), 64 queries, top-40 -- the same synthetic corpus `bench.py` times (bench.make_shard, same seeds).

  * ALL 64 queries against the MFMA-free exact path (`atlas_exact_topk`): ids and score bits;
  * 4 queries against the CPU oracle, the slab streamed through it in 1M-row chunks (the canonical fp16 score of every one of the
    32M rows for each of them, then the canonical top-k): ids and score bits;
  * all 64 queries: the size-independent properties (sorted, no duplicate ids, every returned score is the correctly rounded
    fp64 inner product of the row it names, 8 round-robin shards merged == the single shard).
Skipped when the device cannot hold the slab."""
import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu
N, B, K = 32_000_000, 64, 40


@pytest.fixture(scope="module")
def corpus(gpu_index_cls):
    free, _ = torch.cuda.mem_get_info()
    if free < 70e9:
        pytest.skip("needs ~60 GB of device memory")
    import bench

    dev = torch.device("cuda", 0)
    slab = bench.make_shard(N, 1234, dev)
    q = torch.randn((B, 768), generator=torch.Generator(device=dev).manual_seed(99), device=dev)
    idx = gpu_index_cls()
    idx._set_slab(slab)
    s, i = idx._compute_scores_and_indices(q, K)
    assert idx.last_search_stats["path"] == "scan" and idx.last_search_stats["fallback_queries"] == 0, idx.last_search_stats
    assert idx.last_search_stats["max_err_over_eps"] < 0.25
    return idx, slab, q, s, i


def test_32m_all_64_queries_equal_exact_path(corpus):
    """the whole batch, not a sample (the exact path scores 8 queries per fp64 slab pass: 8 passes of ~20 ms)"""
    idx, slab, q, s, i = corpus
    es, ei = idx._exact_topk(q, K)
    assert torch.equal(s, es) and torch.equal(i, ei)


def test_32m_four_queries_equal_streamed_oracle(corpus, oracle_mod):
    """the slab streamed through the CPU oracle in 1M-row chunks: every row widened to double once and scored against the 4 queries
    (`oracle_search`'s block loop, all N scores kept), then the oracle's canonical top-k of each full score row"""
    idx, slab, q, s, i = corpus
    sel = [7, 31, 40, 63]
    q16 = q[sel].half().cpu().numpy()
    full = np.empty((len(sel), N), dtype=np.float16)
    for r0 in range(0, N, 1_000_000):
        full[:, r0: r0 + 1_000_000] = oracle_mod.search(q16, slab[r0: r0 + 1_000_000].cpu().numpy(), 1, return_full=True)[2]
    for j, b in enumerate(sel):
        es, ei = oracle_mod.topk_row(full[j], K)
        parity.assert_identical(s[b: b + 1].cpu().numpy(), i[b: b + 1].cpu().numpy(), es[None], ei[None], f"32M oracle, query {b}")


def test_32m_properties_all_queries(corpus, gpu_index_cls):
    idx, slab, q, s, i = corpus
    assert (s[:, :-1] >= s[:, 1:]).all()
    assert all(len(set(row)) == K for row in i.tolist())
    assert int(i.min()) >= 0 and int(i.max()) < N
    sub = slab[i.reshape(-1)].double().view(B, K, 768)
    dots = torch.einsum("bkd,bd->bk", sub, q.half().double()).cpu().numpy()
    assert np.array_equal(dots.astype(np.float16).view(np.uint16), s.cpu().numpy().view(np.uint16))   # numpy: one rounding
    # sharding invariance at the 8-GPU shard size: 8 round-robin shards of 4M rows, packed and merged on the device
    from atlas_amd import index as im

    W = 8
    packed = []
    for r in range(W):
        sh = gpu_index_cls()
        sh._set_slab(slab[r::W].contiguous())
        ss, ii = sh._compute_scores_and_indices(q, K)
        packed.append(im.pack_candidates_host(ss.cpu().numpy(), ii.cpu().numpy(), W, r))
        del sh
    merged = idx._merge(torch.from_numpy(np.stack(packed)).cuda(), K)
    ms, mg = im.unpack_candidates_host(merged)
    parity.assert_identical(ms, mg, s.cpu().numpy(), i.cpu().numpy(), "8 x 4M shards vs 32M")
