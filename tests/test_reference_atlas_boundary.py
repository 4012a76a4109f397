"""The drop-in boundary proven with the REFERENCE'S OWN classes (build container only: needs /root/reference; skipped on the GPU box).

1. `src/atlas.py` is imported UNMODIFIED (its only project imports are `src.dist_utils` -- the reference's own file -- and
   `src.retrievers.EMBEDDINGS_DIM`, which resolves to `atlas_amd.retrievers`) and the real `Atlas` class runs
   `build_index` -> `_retrieve` (plain and with a task filter) -> `retrieve_with_rerank` against `HipDistributedIndex` and
   the `atlas_amd.retrievers` wrappers. There is no GPU here, so the two device back-ends are the test stand-ins: the shard-local
   top-k comes from the CPU oracle (tests/oracle_backend.py) and the encoder inside the wrapper is the pinned torch restatement
   (oracle/contriever_ref.py); everything between them and atlas.py -- the (d, N) view written by `embeddings[:, a:b] = emb.T`,
   `search_knn`'s types, the wrapper dispatch, `deepcopy().half().eval()` -- is the product's host code.
2. The on-disk index format against the reference's own `DistributedIndex` (imported through the shim of
   tests/golden/make_golden.py): files saved by `HipDistributedIndex` are loaded by the reference class and vice versa, for the
   128-shard layout of the released indices re-loaded at W = 1, 2, 4, 8, with `W | total_saved_shards` enforced by both.
"""
import logging
import os
import sys
import types

import numpy as np
import pytest
import torch

import synth

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "atlas.py")), reason="reference checkout not present")
HERE = os.path.dirname(os.path.abspath(__file__))


# stand-ins for what is not on this machine: the tokenizer (tests/stub_tokenizer.py: no vocab files offline) and the reader
from stub_tokenizer import HashTokenizer  # noqa: E402


@pytest.fixture
def reference_atlas(monkeypatch):
    """the reference's src/atlas.py, unmodified, with src.retrievers resolving to this package's module"""
    from atlas_amd import retrievers as R

    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setitem(sys.modules, "src.retrievers", R)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)          # atlas.py's _to_cuda / .cuda() on a CPU-only box
    import importlib

    mod = importlib.import_module("src.atlas")
    assert mod.__file__.startswith(REF) and sys.modules["src.dist_utils"].__file__.startswith(REF)
    yield mod
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        del sys.modules[name]


def _make_atlas(mod, monkeypatch, n_passages=37):
    sys.path.insert(0, os.path.dirname(HERE))
    from atlas_amd import HipDistributedIndex, retrievers as R
    from oracle.contriever_ref import BertConfigLite, ContrieverRef
    from oracle_backend import oracle_local_topk

    monkeypatch.setattr(HipDistributedIndex, "_local_topk", oracle_local_topk)
    opt = types.SimpleNamespace(retriever_format="{title} {text}", text_maxlength=24, filtering_overretrieve_ratio=2,
                                n_to_rerank_with_retrieve_with_rerank=12, per_gpu_embedder_batch_size=5, retrieve_with_rerank=False,
                                query_side_retriever_training=False)
    encoder = ContrieverRef(BertConfigLite(vocab_size=1000, num_hidden_layers=2, max_position_embeddings=64), seed=3).randomize_affine()
    retriever = R.DualEncoderRetriever(opt, encoder)
    reader = torch.nn.Linear(1, 1)
    reader_tok = types.SimpleNamespace(vocab={"a": 0, "b": 1})
    tok = HashTokenizer()
    atlas = mod.Atlas(opt, reader, retriever, reader_tok, tok)
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu", "nu", "xi"]
    rng = np.random.default_rng(5)
    passages = [{"id": str(i), "title": f"title {words[i % len(words)]}",
                 "text": " ".join(rng.choice(words, size=int(rng.integers(3, 30))))} for i in range(n_passages)]
    index = HipDistributedIndex()
    index.is_in_gpu = False
    index.init_embeddings(passages)
    return atlas, index, passages, tok, opt


def test_unmodified_atlas_build_index_and_retrieve(reference_atlas, monkeypatch, oracle_mod):
    atlas, index, passages, tok, opt = _make_atlas(reference_atlas, monkeypatch)
    atlas.build_index(index, passages, 8, logger=logging.getLogger("t"))
    # the sic bound of atlas.py:74: max_length = min(text_maxlength, gpu_embedder_batch_size) = 8 tokens, padding 'longest'
    assert [c["n"] for c in tok.calls] == [8, 8, 8, 8, 5] and all(c["max_length"] == 8 and c["padding"] == "longest" for c in tok.calls)
    # what the loop must have produced: the fp16 copy of the retriever on every batch, rows in passage order
    enc16 = atlas._get_fp16_retriever_copy()
    want = torch.cat([enc16(**tok([opt.retriever_format.format(**p) for p in passages[a : a + 8]], padding="longest", return_tensors="pt",
                                  max_length=8, truncation=True), is_passages=True) for a in range(0, len(passages), 8)])
    assert index._slab.dtype == torch.float16 and tuple(index.embeddings.shape) == (768, len(passages))
    assert torch.equal(index._slab, want) and index._slab.is_contiguous()

    query = ["alpha beta", "who is gamma delta", "zeta"]
    qtok = atlas.retriever_tokenize(query)                                           # atlas.py:184-198 (padding='max_length')
    assert tuple(qtok["input_ids"].shape) == (3, opt.text_maxlength)
    stats = {}
    docs, scores, q_emb = atlas._retrieve(index, 6, query, qtok["input_ids"], qtok["attention_mask"], iter_stats=stats)
    assert "runtime/search" in stats and q_emb.dtype == torch.float32 and tuple(q_emb.shape) == (3, 768)
    es, ei = oracle_mod.search(q_emb.half().numpy(), index._slab.numpy(), 6)
    assert [[d["id"] for d in row] for row in docs] == [[str(j) for j in row] for row in ei.tolist()]
    assert scores == es.astype(np.float32).tolist() and docs[0][0] is passages[ei[0, 0]]

    # with a task filter atlas.py:111-113 over-retrieves topk * ratio and filters down
    seen = {}

    def filtering_fun(batch_metadata, passages_, scores_, topk, training):
        seen["k"] = len(passages_[0])
        return [p[1 : topk + 1] for p in passages_], [s[1 : topk + 1] for s in scores_]

    docs_f, scores_f, _ = atlas._retrieve(index, 4, query, qtok["input_ids"], qtok["attention_mask"], batch_metadata=[{}] * 3,
                                          filtering_fun=filtering_fun)
    assert seen["k"] == 8 and [d["id"] for d in docs_f[1]] == [d["id"] for d in docs[1][1:5]]

    # empty local batch (atlas.py:105-106 builds an empty query embedding): search_knn still answers
    e_docs, e_scores, _ = atlas._retrieve(index, 4, [], qtok["input_ids"][:0], qtok["attention_mask"][:0])
    assert e_docs == [] and e_scores == []


def test_unmodified_atlas_retrieve_with_rerank(reference_atlas, monkeypatch):
    atlas, index, passages, tok, opt = _make_atlas(reference_atlas, monkeypatch)
    atlas.build_index(index, passages, 16, logger=logging.getLogger("t"))
    query = ["alpha beta gamma", "kappa lambda"]
    qtok = atlas.retriever_tokenize(query)
    opt.retrieve_with_rerank = True
    docs, scores = atlas.retrieve(index, 5, query, qtok["input_ids"], qtok["attention_mask"])     # atlas.py:120-182
    assert len(docs) == 2 and all(len(r) == 5 for r in docs) and all(len(r) == 5 for r in scores)
    assert all(s[j] >= s[j + 1] for s in scores for j in range(4))
    # the reranked passages are drawn from the n_to_rerank candidates of the first-stage search, re-embedded untruncated
    first, _, _ = atlas._retrieve(index, opt.n_to_rerank_with_retrieve_with_rerank, query, qtok["input_ids"], qtok["attention_mask"])
    for got, cand in zip(docs, first):
        assert {d["id"] for d in got} <= {d["id"] for d in cand}
    rerank_calls = [c for c in tok.calls if c["padding"] == "longest" and c["max_length"] == opt.text_maxlength]
    assert sum(c["n"] for c in rerank_calls) == 2 * opt.n_to_rerank_with_retrieve_with_rerank


# ------------------------------------------------------------------------------------------------------------------
# on-disk format, both directions, against the reference's own DistributedIndex
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def reference_index_cls(monkeypatch):
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden

    cls = make_golden.import_reference_index()           # (leaves sys.modules as it found it; the class keeps its modules alive)
    global _REF_DIST_UTILS
    _REF_DIST_UTILS = cls.__init__.__globals__["dist_utils"]      # the reference's own src/dist_utils.py, as src/index.py imported it
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    yield cls


_REF_DIST_UTILS = None


def _as_rank(monkeypatch, rank, world):
    """both classes ask their dist_utils for the rank / world size: pose as rank `rank` of `world`"""
    import atlas_amd.dist_utils as mine

    ref = _REF_DIST_UTILS
    for m in (mine, ref):
        monkeypatch.setattr(m, "get_rank", lambda r=rank: r)
        monkeypatch.setattr(m, "get_world_size", lambda w=world: w)
    monkeypatch.setattr(mine, "all_gather_object", lambda obj, w=world: [obj] * w)       # (sizes of the other ranks: not under test here)


def _hip_index(P, passages):
    from atlas_amd import HipDistributedIndex

    idx = HipDistributedIndex()
    idx.is_in_gpu = False
    idx.init_embeddings(passages)
    idx.embeddings[:, :] = torch.from_numpy(P).T
    return idx


@pytest.mark.parametrize("N", [1024, 1000])          # 1000: the last files of the 128 hold fewer rows / nothing
def test_reference_class_loads_what_this_class_saved(N, reference_index_cls, monkeypatch, tmp_path):
    P = synth.passages_f16(N, 768, 77)
    passages = [{"id": str(i), "title": f"t{i}", "text": f"passage {i}"} for i in range(N)]
    _as_rank(monkeypatch, 0, 1)
    _hip_index(P, passages).save_index(str(tmp_path), 128)                           # the released indices' shard count
    assert len([f for f in os.listdir(tmp_path) if f.startswith("embeddings.")]) == 128
    for W in (1, 2, 4, 8):
        got_rows, got_docs = [], []
        for rank in range(W):
            _as_rank(monkeypatch, rank, W)
            ref = reference_index_cls()
            ref.is_in_gpu = False
            ref.load_index(str(tmp_path), 128)                                           # src/index.py:89-111, unmodified
            assert ref.embeddings.dtype == torch.float16 and ref.embeddings.shape[0] == 768
            got_rows.append(ref.embeddings.T)
            got_docs += [ref.doc_map[j] for j in range(len(ref.doc_map))]
        assert torch.equal(torch.cat(got_rows), torch.from_numpy(P)) and got_docs == passages, W
    _as_rank(monkeypatch, 0, 3)
    with pytest.raises(AssertionError):
        reference_index_cls().load_index(str(tmp_path), 128)
    from atlas_amd import HipDistributedIndex

    with pytest.raises(AssertionError):
        HipDistributedIndex().load_index(str(tmp_path), 128)


def test_this_class_loads_what_the_reference_class_saved(reference_index_cls, monkeypatch, tmp_path):
    from atlas_amd import HipDistributedIndex

    N, W_save, shards = 1024, 2, 128
    P = synth.passages_f16(N, 768, 78)
    passages = [{"id": str(i), "title": f"t{i}", "text": f"passage {i}"} for i in range(N)]
    for rank in range(W_save):                                                           # a 2-process job of the reference saves 128 shards
        _as_rank(monkeypatch, rank, W_save)
        ref = reference_index_cls()
        ref.is_in_gpu = False
        lo, hi = rank * N // W_save, (rank + 1) * N // W_save
        ref.init_embeddings(passages[lo:hi])
        ref.embeddings[:, :] = torch.from_numpy(P[lo:hi]).T
        ref.save_index(str(tmp_path), shards)                                            # src/index.py:61-87, unmodified
    for W in (1, 2, 4, 8):
        rows, docs = [], []
        for rank in range(W):
            _as_rank(monkeypatch, rank, W)
            idx = HipDistributedIndex()
            idx.is_in_gpu = False
            idx.load_index(str(tmp_path), shards)
            assert idx._slab.is_contiguous() and idx._slab.dtype == torch.float16 and tuple(idx.embeddings.shape) == (768, N // W)
            rows.append(idx._slab)
            docs += [idx.doc_map[j] for j in range(len(idx.doc_map))]
        assert torch.equal(torch.cat(rows), torch.from_numpy(P)) and docs == passages, W
    # and a round trip through this class's own save keeps the reference's bytes: the reference re-loads them
    _as_rank(monkeypatch, 0, 1)
    idx = HipDistributedIndex()
    idx.is_in_gpu = False
    idx.load_index(str(tmp_path), shards)
    out = tmp_path / "again"
    out.mkdir()
    idx.save_index(str(out), shards)
    for s in (0, 63, 127):
        assert torch.equal(torch.load(out / f"embeddings.{s}.pt"), torch.load(tmp_path / f"embeddings.{s}.pt", map_location="cpu"))
