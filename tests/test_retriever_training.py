"""The training forward of the drop-in retriever (atlas_amd/retriever_train.py; reference: src/atlas.py:452-465 over
src/retrievers.py + src/modeling_bert.py). It is torch-operator plumbing around the product (the HIP encoder is an eval-mode,
no-grad encoder), so it runs on the CPU here:

  * against the REFERENCE's own module run live (build container only): train mode, dropout 0.1 on every nn.Dropout, the same
    generator state -> the same dropout masks -> bit-identical embeddings AND bit-identical parameter gradients, with and without
    gradient checkpointing, fp32 and bf16;
  * against the oracle's layers composed under autograd (eval mode; runs everywhere);
  * `set_dropout` of the reference (src/util.py:161-164) finds the dropout modules, `gradient_checkpointing_enable` reaches the
    encoder, the state dict is unchanged by the additions;
  * the retriever step of `Atlas.forward` (atlas.py:457-465) through the wrappers: gradients reach the query encoder, and with
    `query_side_retriever_training` not the passage encoder;
  * which forward goes where: autograd / train-mode dropout -> "autograd"; everything else has to be the HIP encoder, which
    raises on CPU tensors (no eager fallback for inference).
"""
import os
import sys
import types

import pytest
import torch

from atlas_amd import _lib, retrievers as R

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
VOCAB = 613


def _cfg(layers, **kw):
    return R.BertConfigLite(vocab_size=VOCAB, num_hidden_layers=layers, **kw)


def _batch(n, L, seed, holes=False):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, VOCAB, (n, L), generator=g)
    lens = torch.randint(1, L + 1, (n,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    if holes:
        mask = mask * (torch.rand((n, L), generator=g) > 0.2).long()
        mask[:, 0] = 1
    tt = (torch.rand((n, L), generator=g) < 0.3).long() * mask
    return ids, mask, tt


def _randomize(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "LayerNorm" in name or name.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return model


def _loss(emb, seed):
    w = torch.randn(emb.shape, generator=torch.Generator().manual_seed(seed)).to(emb.dtype)
    return (emb * w).sum()


def _reference_set_dropout():
    """src/util.py:161-164, imported from the reference checkout; sys.modules / sys.path are left as they were found (tests/conftest.py checks)"""
    before = set(sys.modules)
    sys.path.insert(0, REF)
    try:
        from src.util import set_dropout
    finally:
        sys.path.remove(REF)
        for name in [m for m in sys.modules if m not in before and (m == "src" or m.startswith("src."))]:
            del sys.modules[name]
    return set_dropout


@pytest.fixture(scope="module")
def ref_mod():
    if not os.path.exists(os.path.join(REF, "src", "retrievers.py")):
        pytest.skip("reference checkout not present")
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden_encoder as mg

    return mg, mg.import_reference_contriever()


@pytest.mark.parametrize("layers,n,L,dtype,ckpt,pooling", [
    (2, 5, 24, torch.float32, False, "average"),
    (2, 4, 33, torch.float32, True, "average"),
    (1, 3, 16, torch.bfloat16, False, "average"),
    (1, 4, 20, torch.float32, False, "sqrt"),
    (1, 4, 20, torch.float32, False, "cls"),
])
def test_train_mode_forward_and_gradients_equal_the_reference_module_bit_for_bit(layers, n, L, dtype, ckpt, pooling, ref_mod):
    from transformers.models.bert.configuration_bert import BertConfig

    mg, ref = ref_mod
    config = BertConfig(vocab_size=VOCAB, hidden_size=768, num_hidden_layers=layers, num_attention_heads=12, intermediate_size=3072,
                        max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                        attention_probs_dropout_prob=0.1)
    theirs = mg.bind_4_18(_randomize(ref.Contriever(config, pooling=pooling), 5)).to(dtype).train()
    mine = R.Contriever(_cfg(layers), pooling=pooling)
    mine.load_state_dict(theirs.state_dict(), strict=True)
    mine = mine.to(dtype).train()
    set_dropout = _reference_set_dropout()                   # the reference's own helper has to find our dropout modules

    set_dropout(theirs, 0.15)
    set_dropout(mine, 0.15)
    assert sum(isinstance(m, torch.nn.Dropout) for m in mine.modules()) == sum(isinstance(m, torch.nn.Dropout) for m in theirs.modules())
    if ckpt:
        theirs.encoder.gradient_checkpointing = True        # what transformers' gradient_checkpointing_enable() sets (modeling_bert.py:559)
        mine.gradient_checkpointing_enable()
    ids, mask, tt = _batch(n, L, 11, holes=True)
    torch.manual_seed(1234)
    want = theirs(input_ids=ids, attention_mask=mask, token_type_ids=tt)
    _loss(want, 3).backward()
    torch.manual_seed(1234)
    got = mine(input_ids=ids, attention_mask=mask, token_type_ids=tt)
    _loss(got, 3).backward()
    assert mine.last_path == "autograd" and got.requires_grad and got.dtype == want.dtype
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    theirs_grads = dict(theirs.named_parameters())
    n_checked = 0
    for name, p in mine.named_parameters():
        g_ref = theirs_grads[name].grad
        assert (p.grad is None) == (g_ref is None), name
        if g_ref is not None:
            assert torch.equal(p.grad, g_ref), (name, float((p.grad.float() - g_ref.float()).abs().max()))
            n_checked += 1
    assert n_checked >= 16 * layers + 5
    # dropout really was active: the eval-mode result differs
    with torch.no_grad():
        assert not torch.equal(theirs.eval()(input_ids=ids, attention_mask=mask, token_type_ids=tt), want)


@pytest.mark.parametrize("layers,n,L", [(1, 4, 12), (2, 3, 31)])
def test_eval_mode_autograd_equals_the_oracle_layers_under_autograd(layers, n, L):
    from oracle.contriever_ref import BertConfigLite, ContrieverRef

    ref = ContrieverRef(BertConfigLite(vocab_size=VOCAB, num_hidden_layers=layers)).randomize_affine().eval()
    mine = R.Contriever(_cfg(layers))
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.eval()
    ids, mask, tt = _batch(n, L, 21)
    got = mine(ids, mask, token_type_ids=tt)
    assert mine.last_path == "autograd" and got.requires_grad
    _loss(got, 8).backward()
    ext = (1.0 - mask[:, None, None, :].float()) * -10000.0                   # the oracle's forward() is no_grad: compose its layers here
    hid = ref.encoder(ref.embeddings(ids, tt), ext)
    hid = hid.masked_fill(~mask[..., None].bool(), 0.0)
    want = hid.sum(dim=1) / mask.sum(dim=1)[..., None]
    _loss(want, 8).backward()
    assert torch.equal(got, want)
    refp = dict(ref.named_parameters())
    for name, p in mine.named_parameters():
        assert torch.equal(p.grad, refp[name].grad), name


def test_additions_do_not_change_the_state_dict_and_checkpointing_flag_is_reachable():
    m = R.Contriever(_cfg(1))
    keys = set(m.state_dict().keys())
    assert not any("dropout" in k for k in keys) and "embeddings.position_ids" in keys and len(keys) == 5 + 16 + 1
    r = R.DualEncoderRetriever(types.SimpleNamespace(), m)
    r.gradient_checkpointing_enable()                                          # retrievers.py:81-83
    assert m.encoder.gradient_checkpointing is True
    r.gradient_checkpointing_disable()
    assert m.encoder.gradient_checkpointing is False
    ps = sorted({mod.p for mod in m.modules() if isinstance(mod, torch.nn.Dropout)})
    assert ps == [0.1] and sum(isinstance(mod, torch.nn.Dropout) for mod in m.modules()) == 1 + 3


@pytest.mark.parametrize("query_side", [False, True])
def test_retriever_step_of_atlas_forward_through_the_wrappers(query_side):
    """atlas.py:457-465: query_emb = retriever(**query_enc, is_passages=False); passage_emb = retriever(**tokens, is_passages=True);
    score = einsum('id,ijd->ij'); the loss reaches the query encoder, and the passage encoder unless it is frozen"""
    opt = types.SimpleNamespace(query_side_retriever_training=query_side)
    q_enc, p_enc = _randomize(R.Contriever(_cfg(1)), 1), _randomize(R.Contriever(_cfg(1)), 2)
    retriever = R.UntiedDualEncoderRetriever(opt, q_enc, p_enc).train()
    for mod in retriever.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                                                        # --dropout 0: the frozen passage side may then use autograd-free torch ops
    bsz, n_ctx = 2, 3
    qi, qm, _ = _batch(bsz, 9, 31)
    pi, pm, _ = _batch(bsz * n_ctx, 14, 32)
    query_emb = retriever(input_ids=qi, attention_mask=qm, is_passages=False)
    if query_side:
        # the frozen passage encoder runs in eval mode without autograd (retrievers.py:126-131): that is an INFERENCE forward, and
        # inference is HIP-only -- on CPU tensors it refuses instead of falling back
        with pytest.raises(_lib.AtlasHipError, match="MI355X|CPU"):
            retriever(input_ids=pi, attention_mask=pm, is_passages=True)
        assert retriever.training and p_enc.training                           # modes restored by the wrapper
        passage_emb = torch.randn(bsz * n_ctx, 768)
    else:
        passage_emb = retriever(input_ids=pi, attention_mask=pm, is_passages=True).to(query_emb)
    passage_emb = passage_emb.view(bsz, -1, passage_emb.size(-1))
    score = torch.einsum("id, ijd->ij", [query_emb, passage_emb])
    torch.nn.functional.log_softmax(score, dim=-1)[:, 0].sum().backward()
    assert all(p.grad is not None and bool(p.grad.abs().sum() > 0) for n, p in q_enc.named_parameters() if "position" not in n and "token_type" not in n)
    if query_side:
        assert all(p.grad is None for p in p_enc.parameters())
    else:
        assert p_enc.embeddings.word_embeddings.weight.grad is not None


def test_inference_never_takes_the_autograd_path():
    m = _randomize(R.Contriever(_cfg(1)), 4)
    ids, mask, _ = _batch(2, 8, 41)
    # eval + no_grad, and eval + parameters that do not require grad: the HIP encoder or nothing
    with torch.no_grad(), pytest.raises(_lib.AtlasHipError):
        m.eval()(ids, mask)
    with pytest.raises(_lib.AtlasHipError):
        m.eval().requires_grad_(False)(ids, mask)
    assert m.last_path == "hip"
    # train mode, no grad, dropout 0: still inference
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    with torch.no_grad(), pytest.raises(_lib.AtlasHipError):
        m.train()(ids, mask)
    # embed_into is the inference entry point only
    out = torch.empty(2, 768)
    with pytest.raises(_lib.AtlasHipError):
        m.requires_grad_(True).embed_into(out, ids, mask)
