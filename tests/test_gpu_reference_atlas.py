"""The reference's OWN `Atlas` class (src/atlas.py, imported unmodified) on the MI355X with BOTH HIP back-ends under it: the HIP Contriever
inside `atlas_amd.retrievers.DualEncoderRetriever` and `HipDistributedIndex` (BASELINE configs[4] in miniature).

Needs a reference checkout: `$ATLAS_REFERENCE_DIR` (a directory holding `src/atlas.py`, `src/dist_utils.py`, `src/slurm.py`), else
`/root/reference`, else `<repo>/.refstage` -- a git-ignored scratch copy that `scripts/stage_reference.sh` makes in the build container so that
one `gpurun` session can carry it to the GPU box (the reference's sources are never committed). Skipped when none is there.

`build_index` -> `_retrieve` (plain and over-retrieving with a task filter) -> `retrieve_with_rerank` run as atlas.py writes them; the search
results are held to the CPU oracle on the slab the HIP encoder wrote and the query embeddings it produced (bit-exact ids and scores), and the
slab itself to the same encoder called directly. No tokenizer vocabulary / checkpoint exists offline: tests/stub_tokenizer.py and random-init
weights stand in for them."""
import importlib
import logging
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _reference_dir():
    for cand in (os.environ.get("ATLAS_REFERENCE_DIR"), "/root/reference", os.path.join(ROOT, ".refstage")):
        if cand and os.path.exists(os.path.join(cand, "src", "atlas.py")):
            return cand
    return None


REF = _reference_dir()


@pytest.fixture
def reference_atlas(monkeypatch):
    """src/atlas.py, unmodified; `src.retrievers` resolves to this package's module (atlas.py only takes EMBEDDINGS_DIM from it)"""
    if REF is None:
        pytest.skip("no reference checkout (ATLAS_REFERENCE_DIR / /root/reference / .refstage)")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from atlas_amd import retrievers as R

    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setitem(sys.modules, "src.retrievers", R)
    mod = importlib.import_module("src.atlas")
    assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REF))
    yield mod
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        del sys.modules[name]


def _make(mod, n_passages=3000, text_maxlength=48):
    from stub_tokenizer import HashTokenizer

    from atlas_amd import HipDistributedIndex, retrievers as R

    opt = types.SimpleNamespace(retriever_format="{title} {text}", text_maxlength=text_maxlength, filtering_overretrieve_ratio=2,
                                n_to_rerank_with_retrieve_with_rerank=24, per_gpu_embedder_batch_size=64, retrieve_with_rerank=False,
                                query_side_retriever_training=False)
    torch.manual_seed(7)
    encoder = R.Contriever(R.BertConfigLite(vocab_size=1000, num_hidden_layers=3, max_position_embeddings=128))
    retriever = R.DualEncoderRetriever(opt, encoder).cuda()                       # model precision fp32, as `--precision fp32`
    reader = torch.nn.Linear(1, 1)
    atlas = mod.Atlas(opt, reader, retriever, types.SimpleNamespace(vocab={"a": 0, "b": 1}), HashTokenizer())
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu", "nu", "xi", "omicron", "pi",
             "rho", "sigma", "tau", "upsilon", "phi", "chi", "psi", "omega"]
    rng = np.random.default_rng(5)
    passages = [{"id": str(i), "title": f"title {words[i % len(words)]} {i}",
                 "text": " ".join(rng.choice(words, size=int(rng.integers(3, 40))))} for i in range(n_passages)]
    index = HipDistributedIndex()
    index.init_embeddings(passages)
    assert index._slab.is_cuda
    return atlas, index, passages, opt


def test_unmodified_atlas_on_the_gpu_build_index_and_retrieve(reference_atlas, oracle_mod):
    import parity

    atlas, index, passages, opt = _make(reference_atlas)
    bs = 512
    atlas.build_index(index, passages, bs, logger=logging.getLogger("t"))           # atlas.py:61-88, unchanged
    tok = atlas.retriever_tokenizer
    # the encoder that served it was the HIP one (the fp16 inference copy of atlas.py:59 is gone; the same copy again, called directly:
    # the slab build_index wrote through `index.embeddings[:, a:b] = emb.T` equals the rows the HIP encoder writes itself)
    enc16 = atlas._get_fp16_retriever_copy()
    want = torch.empty_like(index._slab)
    for a in range(0, len(passages), bs):
        enc = tok([opt.retriever_format.format(**p) for p in passages[a : a + bs]], padding="longest", return_tensors="pt",
                  max_length=min(opt.text_maxlength, bs), truncation=True)
        with torch.no_grad():                                                         # (build_index is @torch.no_grad(), atlas.py:61)
            want[a : a + bs] = enc16(**{k: v.cuda() for k, v in enc.items()}, is_passages=True)
    assert enc16.contriever.last_path == "hip"
    assert index._slab.dtype == torch.float16 and tuple(index.embeddings.shape) == (768, len(passages)) and torch.equal(index._slab, want)

    query = ["alpha beta", "who is gamma delta", "zeta", "omega psi chi phi", "title kappa 7"]
    qtok = atlas.retriever_tokenize(query)                                            # atlas.py:184-198 (padding='max_length')
    stats = {}
    docs, scores, q_emb = atlas._retrieve(index, 10, query, qtok["input_ids"].cuda(), qtok["attention_mask"].cuda(), iter_stats=stats)
    assert atlas.retriever.contriever.last_path == "hip" and index.last_search_stats["path"] == "scan", index.last_search_stats
    assert "runtime/search" in stats and q_emb.dtype == torch.float32 and q_emb.is_cuda and tuple(q_emb.shape) == (5, 768)
    es, ei = oracle_mod.search(q_emb.half().cpu().numpy(), index._slab.cpu().numpy(), 10)
    got_i = np.array([[int(d["id"]) for d in row] for row in docs])
    parity.assert_identical(np.array(scores, dtype=np.float64).astype(np.float16), got_i, es, ei, "Atlas._retrieve on the GPU")
    assert docs[0][0] is passages[int(ei[0, 0])]
    print("reference Atlas on the GPU: encoder", atlas.retriever.contriever.last_path, "search", index.last_search_stats)

    seen = {}

    def filtering_fun(batch_metadata, passages_, scores_, topk, training):            # a task filter: atlas.py:111-113 over-retrieves topk * ratio
        seen["k"] = len(passages_[0])
        return [p[1 : topk + 1] for p in passages_], [s[1 : topk + 1] for s in scores_]

    docs_f, _, _ = atlas._retrieve(index, 4, query, qtok["input_ids"].cuda(), qtok["attention_mask"].cuda(), batch_metadata=[{}] * 5, filtering_fun=filtering_fun)
    assert seen["k"] == 8 and [d["id"] for d in docs_f[1]] == [d["id"] for d in docs[1][1:5]]
    e_docs, e_scores, _ = atlas._retrieve(index, 4, [], qtok["input_ids"][:0].cuda(), qtok["attention_mask"][:0].cuda())   # atlas.py:105-106
    assert e_docs == [] and e_scores == []


def test_unmodified_atlas_on_the_gpu_retrieve_with_rerank(reference_atlas):
    atlas, index, passages, opt = _make(reference_atlas, n_passages=1500)
    atlas.build_index(index, passages, 256, logger=logging.getLogger("t"))
    query = ["alpha beta gamma", "kappa lambda", "tau 11"]
    qtok = atlas.retriever_tokenize(query)
    opt.retrieve_with_rerank = True
    docs, scores = atlas.retrieve(index, 5, query, qtok["input_ids"].cuda(), qtok["attention_mask"].cuda())     # atlas.py:120-182
    assert len(docs) == 3 and all(len(r) == 5 for r in docs) and all(len(r) == 5 for r in scores)
    assert all(s[j] >= s[j + 1] for s in scores for j in range(4))
    first, _, _ = atlas._retrieve(index, opt.n_to_rerank_with_retrieve_with_rerank, query, qtok["input_ids"].cuda(), qtok["attention_mask"].cuda())
    for got, cand in zip(docs, first):
        assert {d["id"] for d in got} <= {d["id"] for d in cand}
    assert atlas.retriever.contriever.last_path == "hip" and index.last_search_stats["path"] == "scan"
    print("reference Atlas.retrieve_with_rerank on the GPU: encoder", atlas.retriever.contriever.last_path, "search", index.last_search_stats["path"])
