"""The N>1 path on CPU: world_size 2 and 3 over gloo, uneven per-rank batches, round-robin shards.
The shard-local top-k comes from the oracle (tests/oracle_backend.py); everything else is product host code:
query all-gather, packing, the single candidate all-gather, host merge, winners-only passage exchange.
Expected result: identical to the canonical single-shard search over the union (sharding invariance)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, W, port, N, batches, k, out_dir, mode):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from atlas_amd import HipDistributedIndex
    from oracle_backend import oracle_local_topk

    HipDistributedIndex._local_topk = oracle_local_topk
    P = synth.passages_f16(N, 768, 61)
    Qall = synth.queries_f32(sum(batches), 768, 62)
    lo = sum(batches[:rank])
    Q = torch.from_numpy(Qall[lo : lo + batches[rank]])
    idx = HipDistributedIndex()
    idx.is_in_gpu = False
    if mode == "round_robin":
        mine = np.arange(rank, N, W)                       # src/index_io.py:41
        idx.init_embeddings([{"id": str(int(g)), "text": f"p{g}"} for g in mine])
        idx.embeddings[:, :] = torch.from_numpy(P[mine]).T
    else:                                                   # contiguous shards via save/load (index.py:95-99)
        full = HipDistributedIndex()
        full.is_in_gpu = False
        if rank == 0:
            full.init_embeddings([{"id": str(g), "text": f"p{g}"} for g in range(N)])
            full.embeddings[:, :] = torch.from_numpy(P).T
            # pretend a 1-process job saved 2*W shards
            import atlas_amd.dist_utils as du
            r, w = du.get_rank, du.get_world_size
            du.get_rank, du.get_world_size = (lambda: 0), (lambda: 1)
            full.save_index(out_dir, 2 * W)
            du.get_rank, du.get_world_size = r, w
        dist.barrier()
        idx.load_index(out_dir, 2 * W)
    import atlas_amd.dist_utils as du0
    real_exchange, received = du0.exchange_objects, []

    def counting_exchange(per_dst):
        got = real_exchange(per_dst)
        received.append(sum(len(part) for part in got))
        return got

    du0.exchange_objects = counting_exchange
    for rep in range(2):                                    # search_knn is a collective: call it twice
        docs, scores = idx.search_knn(Q, k)
    du0.exchange_objects = real_exchange
    # the text exchange is personalised: a rank receives the k winners of each of ITS OWN queries and nothing else
    # (a passage that wins for several of the rank's queries travels once: only the small batches are sure to have k distinct winners each)
    assert received[0] == received[1] <= batches[rank] * k and (batches[rank] > 8 or received[0] == batches[rank] * k), (received, batches[rank] * k)
    # the same search with a node-local passage store attached: no text collective, same documents
    from atlas_amd.passage_store import PassageStore
    spath = os.path.join(out_dir, "store_" + mode)
    if mode == "round_robin":
        store = PassageStore.open_shared(spath, lambda: ({"id": str(g), "text": f"p{g}"} for g in range(N)))
    else:
        store = PassageStore.open_shared(spath, lambda: PassageStore.iter_saved_index(out_dir, 2 * W))
    assert len(store) == N
    idx.attach_passage_store(store)
    import atlas_amd.dist_utils as du2
    real_gather = du2.all_gather_object
    real_x = du2.exchange_objects
    du2.all_gather_object = lambda obj: (_ for _ in ()).throw(AssertionError("text collective used despite the passage store"))
    du2.exchange_objects = lambda per_dst: (_ for _ in ()).throw(AssertionError("text collective used despite the passage store"))
    docs2, scores2 = idx.search_knn(Q, k)
    du2.all_gather_object, du2.exchange_objects = real_gather, real_x
    assert docs2 == docs and scores2 == scores
    ids = np.array([[int(d["id"]) for d in row] for row in docs], dtype=np.int64).reshape(len(docs), k)
    assert all(d["text"] == f"p{d['id']}" for row in docs for d in row)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids, scores=np.array(scores, dtype=np.float32).reshape(len(docs), k))
    dist.barrier()
    dist.destroy_process_group()


# (70 queries on one rank: beyond the 64 rows of the fixed-size query collective -- every rank sees it in the headers and repeats the call)
@pytest.mark.parametrize("W,batches,mode", [(2, (3, 5), "round_robin"), (3, (4, 0, 2), "round_robin"), (2, (2, 2), "contiguous"),
                                            (2, (70, 1), "round_robin")])
def test_distributed_search_equals_union(W, batches, mode, tmp_path, oracle_mod):
    N, k = 1500, 12
    port = 29600 + W * 7 + len(mode) + sum(batches)
    mp.spawn(_worker, args=(W, port, N, batches, k, str(tmp_path), mode), nprocs=W, join=True)
    P = synth.passages_f16(N, 768, 61)
    Q = synth.queries_f32(sum(batches), 768, 62)
    s, i = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    lo = 0
    for r in range(W):
        got = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        n = batches[r]
        if n:
            assert np.array_equal(got["ids"], i[lo : lo + n]), (r, mode)
            assert np.array_equal(got["scores"], s[lo : lo + n].astype(np.float32))
        else:
            assert got["ids"].size == 0
        lo += n


def _cap_worker(rank, W, port):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import atlas_amd.dist_utils as du

    def cap():
        return next(iter(du._query_caps.values()))[0]

    g = torch.Generator().manual_seed(5 + rank)
    big = torch.randn((130 if rank == 1 else 3, 768), generator=g)
    allq, sizes = du.all_gather_queries(big)                       # one rank beyond the block: every rank repeats the call once
    assert sizes == [3, 130] and tuple(allq.shape) == (133, 768) and cap() == 192
    assert torch.equal(allq[3:], big.half()) if rank == 1 else torch.equal(allq[:3], big.half())
    for i in range(du._CAP_DECAY_CALLS):                           # small batches: the block shrinks back after _CAP_DECAY_CALLS of them
        assert cap() == 192
        small = torch.randn((1 + rank, 768), generator=g)
        allq, sizes = du.all_gather_queries(small)
        assert sizes == [1, 2] and tuple(allq.shape) == (3, 768)
    assert cap() == du._QUERY_CAP_MIN
    allq, sizes = du.all_gather_queries(torch.randn((2, 768), generator=g))
    assert sizes == [2, 2] and cap() == du._QUERY_CAP_MIN
    dist.barrier()
    dist.destroy_process_group()


def test_query_block_grows_and_decays_on_every_rank_alike():
    """the fixed-size query collective's block grows for a large batch and falls back to 64 rows after a run of small ones -- decided
    from the gathered headers, so every rank takes the same decision at the same call"""
    mp.spawn(_cap_worker, args=(2, 29871), nprocs=2, join=True)
