"""Host-side logic of HipDistributedIndex (no GPU): API surface of the reference class, (d,N) view semantics,
packed candidates, save/load in the reference's on-disk format, error behaviour, and the no-fallback rule."""
import os
import pickle
import types

import numpy as np
import pytest
import torch

import parity
import synth
from atlas_amd import HipDistributedIndex, _lib, index as index_mod
from atlas_amd.index_io import load_or_initialize_index, load_passages, save_embeddings_and_index
from oracle_backend import oracle_local_topk


def _mk(n, seed=1, dim=768):
    idx = HipDistributedIndex()
    idx.is_in_gpu = False
    passages = [{"id": str(i), "title": f"t{i}", "text": f"passage {i}"} for i in range(n)]
    idx.init_embeddings(passages, dim)
    P = synth.passages_f16(n, dim, seed)
    return idx, passages, P


def test_reference_surface():
    idx, passages, P = _mk(100)
    assert tuple(idx.embeddings.shape) == (768, 100) and idx.embeddings.dtype == torch.float16
    assert idx.doc_map[7]["id"] == "7" and idx.is_index_trained() is True
    for name in ("init_embeddings", "search_knn", "save_index", "load_index", "train_index",
                 "_compute_scores_and_indices", "_get_saved_embedding_path", "_get_saved_passages_path"):
        assert callable(getattr(idx, name))


def test_embeddings_view_is_the_slab():
    """atlas.py:79 writes `index.embeddings[:, a:b] = emb.T`; those must be contiguous slab rows."""
    idx, _, P = _mk(64)
    emb = torch.from_numpy(P[10:30])
    idx.embeddings[:, 10:30] = emb.T
    assert idx._slab.is_contiguous() and idx._slab.shape == (64, 768)
    assert torch.equal(idx._slab[10:30], emb) and float(idx._slab[:10].abs().sum()) == 0.0
    assert idx.embeddings.data_ptr() == idx._slab.data_ptr()


def test_no_cpu_fallback():
    idx, _, P = _mk(32)
    idx.embeddings[:, :] = torch.from_numpy(P).T
    with pytest.raises(_lib.AtlasHipError, match="no .*CPU fallback|CPU"):
        idx.search_knn(torch.randn(2, 768), 4)
    with pytest.raises(_lib.AtlasHipError):
        idx.slab_pmax()


def test_topk_larger_than_shard_raises_like_torch_topk(monkeypatch):
    monkeypatch.setattr(HipDistributedIndex, "_local_topk", oracle_local_topk)
    idx, _, P = _mk(8)
    with pytest.raises(RuntimeError, match="out of range"):
        idx.search_knn(torch.randn(1, 768), 9)


def test_pack_unpack_merge_host(oracle_mod):
    rng = np.random.default_rng(0)
    W, B, k = 3, 5, 6
    s = rng.standard_normal((W, B, k)).astype(np.float16)
    s[0, 0, :3] = s[1, 0, :3]                       # cross-shard ties
    s[2, 1, 0] = np.float16(-0.0); s[0, 1, 0] = np.float16(0.0)
    rows = rng.integers(0, 1000, (W, B, k)).astype(np.int64)
    rows[1, 2, 4:] = -1                              # padding
    packed = np.stack([index_mod.pack_candidates_host(s[w], rows[w], W, w) for w in range(W)])
    assert (packed[1, 2, 4:] == 0).all() and (packed >= 0).all()
    us, ug = index_mod.unpack_candidates_host(packed)
    valid = rows >= 0
    gid = rows * W + np.arange(W)[:, None, None]
    assert np.array_equal(ug[valid], gid[valid]) and (ug[~valid] == -1).all()
    assert np.array_equal(parity.f16_ordinal(us[valid]), parity.f16_ordinal(s[valid]))
    merged = index_mod.merge_packed_host(packed, k)
    ms, mg = index_mod.unpack_candidates_host(merged)
    os_, og = oracle_mod.merge(s, np.where(valid, gid, -1))
    assert np.array_equal(mg, og) and np.array_equal(parity.f16_ordinal(ms), parity.f16_ordinal(os_))


def test_pack_matches_device_layout():
    """numpy packing == csrc/common.h pack_candidate (compiled for the host)"""
    import ctypes
    from atlas_amd import build

    H = ctypes.CDLL(build.build_host())
    H.h_pack_candidate.restype = ctypes.c_uint64
    H.h_pack_candidate.argtypes = [ctypes.c_uint16, ctypes.c_uint64]
    rng = np.random.default_rng(1)
    s = rng.standard_normal(200).astype(np.float16)
    rows = rng.integers(0, 2 ** 31, 200).astype(np.int64)
    got = index_mod.pack_candidates_host(s, rows, 8, 3)
    for j in range(200):
        assert int(got[j]) == H.h_pack_candidate(int(s[j:j + 1].view(np.uint16)[0]), int(rows[j] * 8 + 3))


def test_search_single_process_matches_oracle(monkeypatch, oracle_mod):
    monkeypatch.setattr(HipDistributedIndex, "_local_topk", oracle_local_topk)
    idx, passages, P = _mk(500, seed=5)
    idx.embeddings[:, :] = torch.from_numpy(P).T
    Q = synth.queries_f32(6, 768, 6)
    docs, scores = idx.search_knn(torch.from_numpy(Q), 10)
    s, i = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, 10)
    assert [[int(d["id"]) for d in row] for row in docs] == i.tolist()
    assert scores == s.astype(np.float32).tolist() and isinstance(scores[0][0], float)


def test_save_load_reference_format(tmp_path, monkeypatch, oracle_mod):
    monkeypatch.setattr(HipDistributedIndex, "_local_topk", oracle_local_topk)
    idx, passages, P = _mk(103, seed=9)
    idx.embeddings[:, :] = torch.from_numpy(P).T
    idx.save_index(str(tmp_path), 4)
    # files look exactly like the reference's (src/index.py:61-87): (768, n) fp16 + pickled passage lists
    n_per = int(np.ceil(103 / 4))
    for s in range(4):
        e = torch.load(tmp_path / f"embeddings.{s}.pt")
        lo, hi = s * n_per, min(103, (s + 1) * n_per)
        assert e.dtype == torch.float16 and tuple(e.shape) == (768, hi - lo) and e.is_contiguous()
        assert torch.equal(e, torch.from_numpy(P[lo:hi]).T)
        assert pickle.load(open(tmp_path / f"passages.{s}.pt", "rb")) == passages[lo:hi]
    # the reference's own load_index logic on these files (restated): concat along dim=1 gives the (d, N) matrix
    ref_emb = torch.concat([torch.load(tmp_path / f"embeddings.{s}.pt") for s in range(4)], dim=1)
    assert torch.equal(ref_emb, torch.from_numpy(P).T)
    idx2 = HipDistributedIndex()
    idx2.is_in_gpu = False
    idx2.load_index(str(tmp_path), 4)
    assert torch.equal(idx2._slab, torch.from_numpy(P)) and idx2.doc_map == idx.doc_map
    Q = torch.from_numpy(synth.queries_f32(3, 768, 10))
    assert idx2.search_knn(Q, 5) == idx.search_knn(Q, 5)
    # passages are not rewritten unless asked (index.py:80-83)
    before = os.path.getmtime(tmp_path / "passages.0.pt")
    idx.save_index(str(tmp_path), 4)
    assert os.path.getmtime(tmp_path / "passages.0.pt") == before
    with pytest.raises(AssertionError):
        HipDistributedIndex().save_index(str(tmp_path), 4)


def test_index_io_factory(tmp_path):
    f = tmp_path / "p.jsonl"
    f.write_text('{"id": "0", "title": "A", "section": "s", "text": "x"}\n{"id": "1", "title": "B", "text": "y"}\n')
    assert load_passages([str(f)])[0]["title"] == "A: s"          # index_io.py:30-31
    opt = types.SimpleNamespace(index_mode="flat", load_index_path=None, passages=[str(f)], use_file_passages=False,
                                max_passages=-1)
    from atlas_amd import index as im
    orig = im.HipDistributedIndex._device
    im.HipDistributedIndex._device = lambda self: torch.device("cpu")
    try:
        index, passages = load_or_initialize_index(opt)
    finally:
        im.HipDistributedIndex._device = orig
    assert isinstance(index, HipDistributedIndex) and len(passages) == 2 and tuple(index.embeddings.shape) == (768, 2)
    opt.index_mode = "faiss"
    with pytest.raises(ValueError, match="unsupported index mode"):
        load_or_initialize_index(opt)


def test_passage_store_roundtrip(tmp_path):
    """jsonl -> store -> get: same parsing as load_passages (title/section join, None for blank lines), unicode, atomics"""
    import json
    from atlas_amd.passage_store import PassageStore

    f = tmp_path / "p.jsonl"
    items = [{"id": "0", "title": "A", "section": "s", "text": "x"}, None, {"id": "2", "title": "B", "section": "", "text": "é ü 漢"},
             {"id": "3", "text": "last"}]
    f.write_text("\n".join("" if it is None else json.dumps(it) for it in items) + "\n")
    path = str(tmp_path / "store")
    st = PassageStore.open_shared(path, lambda: PassageStore.iter_jsonl([str(f)]))
    assert len(st) == 4
    assert st.get(0)["title"] == "A: s" and st.get(1) is None and st.get(2)["text"] == "é ü 漢" and st[3] == {"id": "3", "text": "last"}
    st2 = PassageStore.open_shared(path, lambda: (_ for _ in ()).throw(AssertionError("must not rebuild an existing store")))
    assert st2.get(3) == st.get(3)
    st3_path = str(tmp_path / "store_max")
    st3 = PassageStore.open_shared(st3_path, lambda: PassageStore.iter_jsonl([str(f)], maxload=2))
    assert len(st3) == 2


def test_contriever_from_pretrained_local_dir(tmp_path):
    """src/model_io.py:45 loads the retriever with Contriever.from_pretrained(path): HF directory layout, both file formats,
    optional `bert.` prefix and pooler weights"""
    import json
    import torch
    from safetensors.torch import save_file
    from atlas_amd import retrievers

    cfg = dict(vocab_size=50, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=16, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu", model_type="bert")
    src = retrievers.Contriever(retrievers.BertConfigLite(vocab_size=50, num_hidden_layers=1, max_position_embeddings=16))
    sd = {k: v.detach().clone() for k, v in src.state_dict().items()}
    for sub, fmt, prefix in (("a", "safetensors", ""), ("b", "bin", "bert.")):
        d = tmp_path / sub
        d.mkdir()
        (d / "config.json").write_text(json.dumps(cfg))
        out = {prefix + k: v for k, v in sd.items()}
        out[prefix + "pooler.dense.weight"] = torch.zeros(768, 768)
        out[prefix + "embeddings.position_ids"] = torch.arange(16)[None]
        if fmt == "safetensors":
            save_file(out, str(d / "model.safetensors"))
        else:
            torch.save(out, str(d / "pytorch_model.bin"))
        m = retrievers.Contriever.from_pretrained(str(d))
        assert m.config.num_hidden_layers == 1 and m.config.pooling == "average"
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd[k]), k
    (tmp_path / "a" / "config.json").write_text(json.dumps({**cfg, "hidden_act": "relu"}))
    with pytest.raises(Exception, match="hidden_act"):
        retrievers.Contriever.from_pretrained(str(tmp_path / "a"))


def test_doc_map_mirror_follows_in_place_edits_and_re_initialisation(monkeypatch, oracle_mod):
    """the id -> passage mirror of `_docs_of_rows` (a) sees `index.doc_map[i] = p` like the reference, which re-reads doc_map[x]
    on every search (src/index.py:133); (b) is dropped with the slab, so two re-initialisations with no search between never serve the first
    corpus' passages, whatever id() the new dict gets; (c) holds for a caller's own plain dict too (new object, or invalidate_doc_cache)."""
    from oracle_backend import oracle_local_topk

    monkeypatch.setattr(HipDistributedIndex, "_local_topk", oracle_local_topk)
    P = synth.passages_f16(50, 768, 3)
    Q = torch.from_numpy(synth.queries_f32(2, 768, 4))
    idx = HipDistributedIndex()
    idx.is_in_gpu = False
    idx.init_embeddings([{"id": str(i), "text": f"a{i}"} for i in range(50)])
    idx.embeddings[:, :] = torch.from_numpy(P).T
    docs, _ = idx.search_knn(Q, 3)
    top = int(docs[0][0]["id"])
    assert docs[0][0]["text"] == f"a{top}"
    idx.doc_map[top] = {"id": str(top), "text": "edited"}                    # (a) in place, same dict, same length
    assert idx.search_knn(Q, 3)[0][0][0]["text"] == "edited"
    for tag in ("b", "c"):                                                   # (b) two re-inits, no search between
        idx.init_embeddings([{"id": str(i), "text": f"{tag}{i}"} for i in range(50)])
    idx.embeddings[:, :] = torch.from_numpy(P).T
    assert idx.search_knn(Q, 3)[0][0][0]["text"] == f"c{top}"
    idx.doc_map = {i: {"id": str(i), "text": f"d{i}"} for i in range(50)}     # (c) the caller's own plain dict
    assert idx.search_knn(Q, 3)[0][0][0]["text"] == f"d{top}"
    idx.doc_map[top] = {"id": str(top), "text": "plain-edit"}                # a plain dict cannot tell: the documented call
    idx.invalidate_doc_cache()
    assert idx.search_knn(Q, 3)[0][0][0]["text"] == "plain-edit"
    import copy
    import pickle

    assert pickle.loads(pickle.dumps(idx.doc_map)) == idx.doc_map and copy.deepcopy(idx.doc_map) == idx.doc_map


def test_topk_check_is_taken_from_the_smallest_shard(monkeypatch):
    """the range check of search_knn uses the job's smallest shard (one collective per slab, cached): single process = this shard"""
    from oracle_backend import oracle_local_topk

    monkeypatch.setattr(HipDistributedIndex, "_local_topk", oracle_local_topk)
    idx = HipDistributedIndex()
    idx.is_in_gpu = False
    idx.init_embeddings([{"id": str(i)} for i in range(5)])
    idx.embeddings[:, :] = torch.from_numpy(synth.passages_f16(5, 768, 8)).T
    Q = torch.from_numpy(synth.queries_f32(1, 768, 9))
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        idx.search_knn(Q, 6)
    assert len(idx.search_knn(Q, 5)[0][0]) == 5 and idx._min_shard_rows == 5
    idx.init_embeddings([{"id": str(i)} for i in range(9)])                 # a new slab: the bound is taken again
    assert idx._min_shard_rows is None


def test_passage_store_payload_is_pickled_and_old_json_stores_still_read(tmp_path):
    """round 5: the store's payload is one pickle per passage (0.6 us per lookup against 3.3 us for JSON: 2 560 winners per search and rank);
    a store of the older JSON format is still READ entry by entry (first byte), and REBUILT by open_shared when a signature is given"""
    import json
    from atlas_amd.passage_store import PassageStore

    items = [{"id": "0", "title": "A", "text": "x é 漢"}, None, {"id": "2", "text": "y"}]
    new = str(tmp_path / "new")
    PassageStore.build_from_items(new, items)
    st = PassageStore(new)
    assert [st.get(i) for i in range(3)] == items and st.get_many([2, 0, 1, 2]) == [items[2], items[0], None, items[2]]
    assert open(new + ".bin", "rb").read(1) == b"\x80"
    # an old-format store, written by hand
    old = str(tmp_path / "old")
    blobs = [json.dumps(it, ensure_ascii=False).encode() if it is not None else b"" for it in items]
    open(old + ".bin", "wb").write(b"".join(blobs))
    np.save(old + ".off.npy", np.cumsum([0] + [len(b) for b in blobs]).astype(np.int64))
    assert [PassageStore(old).get(i) for i in range(3)] == items and PassageStore(old).get_many([0, 1, 2]) == items
    json.dump({"signature": "s"}, open(old + ".meta.json", "w"))            # (no "format": rounds 2-4)
    rebuilt = PassageStore.open_shared(old, lambda: iter(items), signature="s")
    assert open(old + ".bin", "rb").read(1) == b"\x80" and [rebuilt.get(i) for i in range(3)] == items
    assert json.load(open(old + ".meta.json"))["format"] == 2


def test_passage_store_bounds_and_sequence_protocol(tmp_path):
    """an id outside [0, len) raises IndexError (get and get_many), so iteration over a store terminates and a negative id never wraps
    around to the last passages"""
    from atlas_amd.passage_store import PassageStore

    items = [{"id": str(i), "text": f"t{i}"} for i in range(5)]
    path = str(tmp_path / "s")
    PassageStore.build_from_items(path, items)
    st = PassageStore(path)
    assert list(st) == items and [p["id"] for p in st] == ["0", "1", "2", "3", "4"]
    for bad in (5, -1, 10**9):
        with pytest.raises(IndexError):
            st.get(bad)
    with pytest.raises(IndexError):
        st.get_many([0, -1])
    with pytest.raises(IndexError):
        st.get_many([4, 5])
    assert st.get_many([]) == [] and st.get_many([4, 0]) == [items[4], items[0]]


def test_automatic_passage_store_is_private_to_the_user(tmp_path):
    """an automatic store is only reused when its three files and its directory belong to this user and nobody else can write
    to them; a store that fails the check is never unpickled -- it is replaced by one built from the corpus"""
    import json
    import pickle
    from atlas_amd.passage_store import PassageStore, PassageStoreError

    base = tmp_path / "shm"
    base.mkdir()
    d = PassageStore.private_dir(str(base))
    assert os.path.basename(d) == "atlas_amd_%d" % os.getuid() and (os.stat(d).st_mode & 0o777) == 0o700
    path = os.path.join(d, "passages_abc")
    items = [{"id": "0", "text": "real"}]
    st = PassageStore.open_shared(path, lambda: iter(items), signature="sig", require_private=True)
    assert st.get(0) == items[0] and PassageStore.is_private(path)
    assert all((os.stat(path + e).st_mode & 0o077) == 0 for e in (".bin", ".off.npy", ".meta.json"))
    # a planted store with the right signature and format whose payload is group-writable: rebuilt, its pickle never loaded
    class Boom:
        def __reduce__(self):
            return (pytest.fail, ("the planted pickle was loaded",))
    with open(path + ".bin", "wb") as f:
        f.write(pickle.dumps(Boom(), protocol=5))
    os.chmod(path + ".bin", 0o664)
    assert not PassageStore.is_private(path)
    st2 = PassageStore.open_shared(path, lambda: iter(items), signature="sig", require_private=True)
    assert st2.get(0) == items[0] and PassageStore.is_private(path)
    # a world-writable directory is refused outright
    os.chmod(d, 0o777)
    try:
        with pytest.raises(PassageStoreError):
            PassageStore.private_dir(str(base))
        with pytest.raises(PassageStoreError, match="not trusted"):
            PassageStore.open_shared(path, lambda: iter(items), signature="sig", require_private=True)
    finally:
        os.chmod(d, 0o700)
    # an explicit store (a path the user chose) keeps the reference's trust model: no ownership requirement
    ex = str(tmp_path / "explicit")
    PassageStore.open_shared(ex, lambda: iter(items), signature="sig")
    os.chmod(ex + ".bin", 0o664)
    assert PassageStore.open_shared(ex, lambda: pytest.fail("rebuilt"), signature="sig").get(0) == items[0]


def test_saved_index_signature_sees_a_resave_with_equal_sizes(tmp_path):
    """the signature of a restored index holds every shard pickle's mtime, not only its size"""
    import pickle
    import types
    from atlas_amd import index_io

    for s in range(2):
        with open(tmp_path / f"passages.{s}.pt", "wb") as f:
            pickle.dump([{"id": str(s), "text": "aaaa"}], f)
    opt = types.SimpleNamespace(load_index_path=str(tmp_path), save_index_n_shards=2, passages=[], max_passages=-1)
    a = index_io._corpus_signature(opt)
    assert a == index_io._corpus_signature(opt)
    with open(tmp_path / "passages.1.pt", "wb") as f:
        pickle.dump([{"id": "1", "text": "bbbb"}], f)            # same size, other text
    st = os.stat(tmp_path / "passages.1.pt")
    os.utime(tmp_path / "passages.1.pt", ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))
    assert index_io._corpus_signature(opt) != a


def test_refresh_fingerprint_sees_a_length_changing_edit_of_any_passage():
    """the token store of build_index_streamed is reused only while the sampled entries AND the total text length are unchanged"""
    from atlas_amd.refresh import _passages_fingerprint

    ps = [{"id": str(i), "title": "t", "text": "x" * (10 + i % 7)} for i in range(5000)]
    f0 = _passages_fingerprint(ps)
    assert f0 == _passages_fingerprint(ps) and _passages_fingerprint([]) == (0, 0)
    picks = {0, 4999, *range(0, 5000, 5000 // 62)}
    unsampled = next(i for i in range(5000) if i not in picks)
    ps[unsampled]["text"] += "!"
    assert _passages_fingerprint(ps) != f0
    ps[unsampled]["text"] = ps[unsampled]["text"][:-1]
    assert _passages_fingerprint(ps) == f0
    ps[0]["text"] = "y" + ps[0]["text"][1:]                        # a same-length edit of a sampled entry
    assert _passages_fingerprint(ps) != f0
    assert _passages_fingerprint([{"id": "0"}, {"id": "1", "text": None}]) [1] == 0
