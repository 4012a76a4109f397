"""The reference's call sequence on the MI355X with both halves of the path swapped in:

    Atlas.build_index   (src/atlas.py:52-88):  retriever_fp16 = deepcopy(retriever).half().eval();  per batch
                                               emb = retriever_fp16(**enc, is_passages=True);  index.embeddings[:, a:b] = emb.T
    Atlas._retrieve     (src/atlas.py:90-118): query_emb = retriever(query_ids, query_mask, is_passages=False)  (model precision)
                                               passages, scores = index.search_knn(query_emb, topk)

No tokenizer / checkpoint is available offline: token batches are synthetic and the weights integer-generated
(tests/synth_encoder.py); what is under test is the composition — the embeddings written by the encoder are the rows the
search reads, in the dtypes atlas.py uses. The search result is bit-exact against the CPU oracle run on the slab and
the query embeddings the HIP encoder produced; the encoder itself is held to the reference in test_encoder_golden.py."""
import copy
import types

import numpy as np
import pytest
import torch

import parity
import synth_encoder

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [torch.float32, torch.bfloat16])
def test_build_index_then_retrieve(precision, gpu_index_cls, oracle_mod):
    from atlas_amd import retrievers

    case = {"name": "e2e", "layers": 2, "vocab": 3000, "seed": 21}
    c = synth_encoder.config_dict(case)
    contriever = retrievers.Contriever(retrievers.BertConfigLite(vocab_size=c["vocab_size"], num_hidden_layers=2))
    contriever.load_state_dict(synth_encoder.state_dict(case), strict=True)
    retriever = retrievers.DualEncoderRetriever(types.SimpleNamespace(), contriever).to(precision).cuda()   # model precision

    N, L, bs, k, nq = 1500, 24, 256, 10, 12
    g = torch.Generator().manual_seed(4)
    tok = torch.randint(1000, 3000, (N, L), generator=g)
    lens = torch.randint(6, L + 1, (N,), generator=g)
    pmask = (torch.arange(L)[None, :] < lens[:, None]).long()
    tok = tok * pmask
    passages = [{"id": str(i), "title": f"t{i}", "text": f"passage {i}"} for i in range(N)]

    index = gpu_index_cls()
    index.init_embeddings(passages)
    with torch.no_grad():                                                  # atlas.py:52 @torch.no_grad()
        retriever_fp16 = copy.deepcopy(retriever).half().eval()            # atlas.py:59
        for a in range(0, N, bs):
            b = min(N, a + bs)
            emb = retriever_fp16(input_ids=tok[a:b].cuda(), attention_mask=pmask[a:b].cuda(), is_passages=True)
            index.embeddings[:, a:b] = emb.T                               # atlas.py:79, unchanged
        # the fused form (pooling epilogue writes the slab rows) gives the same slab
        slab_ref = index._slab.clone()
        retriever_fp16.contriever.embed_into(index._slab[0:bs], tok[0:bs].cuda(), pmask[0:bs].cuda())
        assert torch.equal(index._slab, slab_ref)

        # queries: padded to a fixed max_length like retriever_tokenize, model precision
        qlen = torch.randint(4, 12, (nq,), generator=g)
        qtok = torch.randint(1000, 3000, (nq, 64), generator=g)
        qmask = (torch.arange(64)[None, :] < qlen[:, None]).long()
        qtok = qtok * qmask
        retriever.eval()
        query_emb = retriever(qtok.cuda(), qmask.cuda(), is_passages=False)           # atlas.py:104
        assert query_emb.dtype == precision and not torch.isnan(query_emb.float()).any()
        docs, scores = index.search_knn(query_emb, k)                                  # atlas.py:106

    # oracle on exactly what the encoder produced: search_knn casts queries with .half() (index.py:117)
    es, ei = oracle_mod.search(query_emb.half().cpu().numpy(), index._slab.cpu().numpy(), k)
    got_ids = np.array([[int(d["id"]) for d in row] for row in docs])
    got_scores = np.array(scores, dtype=np.float16)
    parity.assert_identical(got_scores, got_ids, es, ei, f"end-to-end {precision}")
    assert docs[0][0]["title"] == f"t{got_ids[0][0]}"


def test_streamed_refresh_equals_batch_loop(gpu_index_cls, oracle_mod):
    """atlas_amd.refresh.IndexRefresher (pinned staging + copy stream + slab-row epilogue) writes the same
    slab as the reference-style loop, batch sizes and lengths varying, and the refreshed index searches exactly"""
    from atlas_amd import refresh, retrievers

    case = {"name": "e2e", "layers": 2, "vocab": 3000, "seed": 22}
    c = synth_encoder.config_dict(case)
    enc = retrievers.Contriever(retrievers.BertConfigLite(vocab_size=c["vocab_size"], num_hidden_layers=2))
    enc.load_state_dict(synth_encoder.state_dict(case), strict=True)
    enc = enc.half().eval().cuda().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    sizes = [(200, 32), (256, 48), (17, 20), (256, 64), (130, 33), (1, 5)]
    N = sum(n for n, _ in sizes) + 40
    batches = []
    for n, L in sizes:
        tok = torch.randint(1000, 3000, (n, L), generator=g)
        lens = torch.randint(3, L + 1, (n,), generator=g)
        m = (torch.arange(L)[None, :] < lens[:, None]).long()
        batches.append((tok * m, m))
    index = gpu_index_cls()
    index.init_embeddings([{"id": str(i)} for i in range(N)])
    r = refresh.IndexRefresher(index, enc, max_batch=256, max_len=64, depth=2)
    wrote = r.run(iter(batches), row_offset=20)
    torch.cuda.synchronize()
    assert wrote == N - 40
    want = torch.zeros_like(index._slab)
    row = 20
    for ids, m in batches:
        want[row: row + ids.shape[0]] = enc(ids.cuda(), m.cuda())
        row += ids.shape[0]
    assert torch.equal(index._slab, want)
    q = torch.randn((5, 768), generator=torch.Generator().manual_seed(6))
    docs, scores = index.search_knn(q.cuda(), 7)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(q.numpy()), want.cpu().numpy(), 7)
    parity.assert_identical(np.array(scores, dtype=np.float16), np.array([[int(d["id"]) for d in row] for row in docs]), es, ei, "after refresh")


def _small_encoder(seed=23):
    from atlas_amd import retrievers

    case = {"name": "e2e", "layers": 2, "vocab": 3000, "seed": seed}
    c = synth_encoder.config_dict(case)
    enc = retrievers.Contriever(retrievers.BertConfigLite(vocab_size=c["vocab_size"], num_hidden_layers=2))
    sd = synth_encoder.state_dict(case)
    enc.load_state_dict(sd, strict=True)
    return enc, sd, c


def test_refresh_from_token_store_equals_position_loop_and_the_restatement(gpu_index_cls):
    """a refresh streamed from the pinned, length-bucketed token store writes the slab the position-ordered batch
    loop writes (bit for bit: a passage's embedding does not depend on its batch mates), twice in a row (weights unchanged), and
    one of its batches is held to the torch restatement of the reference encoder (oracle/contriever_ref.py, fp16 on the GPU)."""
    from atlas_amd import refresh
    from atlas_amd.token_store import TokenStore
    from oracle.contriever_ref import BertConfigLite, ContrieverRef

    enc, sd, c = _small_encoder()
    enc = enc.half().eval().cuda().requires_grad_(False)
    rng = np.random.default_rng(11)
    N, bs = 700, 128
    lists = [[101] + rng.integers(1000, 3000, size=int(rng.integers(2, 70))).tolist() + [102] for _ in range(N)]
    store = TokenStore.from_token_lists(lists, max_length=72)
    index = gpu_index_cls()
    index.init_embeddings([{"id": str(i)} for i in range(N)])
    r = refresh.IndexRefresher(index, enc, max_batch=bs, max_len=72, depth=3)
    assert r.run_store(store, bs) == N
    torch.cuda.synchronize()
    want = torch.empty_like(index._slab)
    for rows, ids, mask in store.batches(bs, bucket=False):                      # the reference's order: by position
        want[rows[0] : rows[-1] + 1] = enc(ids.cuda(), mask.cuda())
    assert torch.equal(index._slab, want)
    index._slab.zero_()
    assert r.run_store(store, bs, repeat=2) == 2 * N                           # back-to-back refreshes reuse the staging slots
    torch.cuda.synchronize()
    assert torch.equal(index._slab, want)
    # one bucketed batch against the restatement of the reference (fp16 weights, torch ops on the GPU): north_star's 1e-3 class
    ref = ContrieverRef(BertConfigLite(vocab_size=c["vocab_size"], num_hidden_layers=2))
    ref.load_state_dict(sd, strict=True)
    ref = ref.half().eval().cuda()
    rows, ids, mask = next(iter(store.batches(bs, bucket=True)))
    exp = ref(ids.cuda(), mask.cuda()).float()
    got = index._slab[torch.from_numpy(rows).cuda()].float()
    err = float((got - exp).abs().max() / exp.abs().max())
    print(f"refreshed batch vs restatement: max|d|/max|e| = {err:.2e}")
    assert err <= 2e-3


def test_build_index_streamed_is_a_drop_in_for_atlas_build_index(gpu_index_cls, oracle_mod):
    """`atlas_amd.refresh.build_index_streamed` bound in place of `Atlas.build_index` (same signature): tokenises once, keeps a
    persistent fp16 mirror of the (fp32, training) retriever, and produces the slab of the reference's loop -- restated here line by
    line as in test_build_index_then_retrieve -- also after the retriever's weights moved (the mirror is re-cast in place)."""
    import logging

    from atlas_amd import refresh, retrievers
    from stub_tokenizer import HashTokenizer

    enc, _, _ = _small_encoder(seed=24)
    retriever = retrievers.DualEncoderRetriever(types.SimpleNamespace(), enc).cuda().train()
    rng = np.random.default_rng(12)
    words = [f"w{j}" for j in range(400)]
    passages = [{"id": str(i), "title": f"t{i % 7}", "text": " ".join(rng.choice(words, size=int(rng.integers(1, 50))))} for i in range(333)]
    opt = types.SimpleNamespace(retriever_format="{title} {text}", text_maxlength=48)
    model = types.SimpleNamespace(retriever=retriever, retriever_tokenizer=HashTokenizer(vocab_size=3000), opt=opt)
    model.build_index = types.MethodType(refresh.build_index_streamed, model)
    bs = 64

    def reference_loop(index):                                                    # atlas.py:61-88, restated
        retrieverfp16 = copy.deepcopy(retriever).half().eval()
        tok, total = HashTokenizer(vocab_size=3000), 0
        with torch.no_grad():
            for i in range(0, len(passages), bs):
                batch = [opt.retriever_format.format(**example) for example in passages[i : i + bs]]
                batch_enc = tok(batch, padding="longest", return_tensors="pt", max_length=min(opt.text_maxlength, bs), truncation=True)
                embeddings = retrieverfp16(**{k: v.cuda() for k, v in batch_enc.items()}, is_passages=True)
                index.embeddings[:, total : total + len(embeddings)] = embeddings.T
                total += len(embeddings)

    index, want = gpu_index_cls(), gpu_index_cls()
    for ix in (index, want):
        ix.init_embeddings(passages)
    for step in range(2):
        model.build_index(index, passages, bs, logger=logging.getLogger("t"))
        reference_loop(want)
        assert torch.equal(index._slab, want._slab), f"refresh {step}"
        assert len(model.retriever_tokenizer.calls) == 1                         # tokenised once, not per refresh
        with torch.no_grad():                                                     # "a training step": the weights move
            for p in retriever.parameters():
                p.mul_(1.0 + 0.01 * (step + 1))
    q = torch.randn((3, 768), generator=torch.Generator().manual_seed(7))
    docs, scores = index.search_knn(q.cuda(), 5)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(q.numpy()), want._slab.cpu().numpy(), 5)
    parity.assert_identical(np.array(scores, dtype=np.float16), np.array([[int(d["id"]) for d in row] for row in docs]), es, ei, "streamed build_index")


def test_save_load_search_and_passage_store_over_rccl(gpu_index_cls, oracle_mod, tmp_path):
    """On the device: save_index -> load_index (each (d, n) block transposed into the pre-allocated slab on the GPU) -> search equals
    the oracle; the same through the distributed branch over RCCL (world size 1) with the node-local passage store attached, i.e.
    with no text collective."""
    import os
    import socket

    import torch.distributed as dist

    import synth
    from atlas_amd import dist_utils
    from atlas_amd.passage_store import PassageStore

    N, k = 3001, 9
    P = synth.passages_f16(N, 768, 97)
    passages = [{"id": str(i), "title": f"t{i}", "text": f"x{i}"} for i in range(N)]
    src = gpu_index_cls()
    src.init_embeddings(passages)
    src.embeddings[:, :] = torch.from_numpy(P).cuda().T
    src.save_index(str(tmp_path), 8)
    idx = gpu_index_cls()
    idx.load_index(str(tmp_path), 8)
    assert idx._slab.is_cuda and idx._slab.is_contiguous() and torch.equal(idx._slab.cpu(), torch.from_numpy(P)) and idx.doc_map == src.doc_map
    Q = synth.queries_f32(6, 768, 98)
    es, ei = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    docs, scores = idx.search_knn(torch.from_numpy(Q).cuda(), k)
    assert [[int(d["id"]) for d in row] for row in docs] == ei.tolist() and scores == es.astype(np.float32).tolist()

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        idx2 = gpu_index_cls()
        idx2.load_index(str(tmp_path), 8)
        store = PassageStore.open_shared(str(tmp_path / "store"), lambda: PassageStore.iter_saved_index(str(tmp_path), 8), signature="t")
        idx2.attach_passage_store(store)
        real, real_x = dist_utils.all_gather_object, dist_utils.exchange_objects
        calls = []
        dist_utils.all_gather_object = lambda obj: calls.append(1) or real(obj)
        dist_utils.exchange_objects = lambda per_dst: calls.append(1) or real_x(per_dst)
        try:
            docs2, scores2 = idx2.search_knn(torch.from_numpy(Q).cuda(), k)
        finally:
            dist_utils.all_gather_object, dist_utils.exchange_objects = real, real_x
        assert not calls, "text collective used despite the passage store"
        assert docs2 == docs and scores2 == scores
    finally:
        dist.destroy_process_group()
