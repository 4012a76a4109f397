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
