"""The retriever wrappers of src/retrievers.py:63-135 around `atlas_amd.retrievers.Contriever` (no GPU: the encoders are
stand-ins that record how they were called) and the checkpoint-key contract of src/model_io.py:62-71, 109-122."""
import copy
import types

import pytest
import torch
import torch.nn as nn

from atlas_amd import retrievers as R


class _Probe(nn.Module):
    """an 'encoder' that reports the mode / autograd state it was run in"""

    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.ones(3))
        self.calls = []
        self.ckpt = None

    def forward(self, x, scale=1.0):
        self.calls.append((self.training, torch.is_grad_enabled()))
        return x * self.w * scale

    def gradient_checkpointing_enable(self):
        self.ckpt = True

    def gradient_checkpointing_disable(self):
        self.ckpt = False


def test_base_retriever_contract():
    base = R.BaseRetriever()
    with pytest.raises(NotImplementedError):
        base.embed_queries(torch.ones(1, 3))
    with pytest.raises(NotImplementedError):
        base(torch.ones(1, 3), is_passages=True)


def test_dual_encoder_shares_one_module():
    enc = _Probe()
    r = R.DualEncoderRetriever(types.SimpleNamespace(), enc)
    x = torch.arange(6.0).view(2, 3)
    assert torch.equal(r(x), x) and torch.equal(r(x, is_passages=True, scale=2.0), 2 * x)
    assert torch.equal(r._embed(x), x) and len(enc.calls) == 3
    assert [k for k, _ in r.state_dict().items()] == ["contriever.w"]
    r.gradient_checkpointing_enable()
    assert enc.ckpt is True
    r.gradient_checkpointing_disable()
    assert enc.ckpt is False


@pytest.mark.parametrize("train_mode", [True, False])
def test_untied_freezes_the_passage_side_for_query_side_training(train_mode):
    """retrievers.py:123-135: eval + no_grad around the passage encoder, mode restored afterwards"""
    q, p = _Probe(), _Probe()
    r = R.UntiedDualEncoderRetriever(types.SimpleNamespace(query_side_retriever_training=True), q, p)
    r.train(train_mode)
    x = torch.ones(2, 3)
    out_p = r(x, is_passages=True)
    out_q = r(x)
    assert p.calls == [(False, False)] and p.training is train_mode          # ran frozen, mode put back
    assert q.calls == [(train_mode, True)]
    assert not out_p.requires_grad and out_q.requires_grad
    r.opt.query_side_retriever_training = False
    assert r(x, is_passages=True).requires_grad and p.calls[-1] == (train_mode, True)
    assert sorted(r.state_dict()) == ["passage_contriever.w", "query_contriever.w"]


def test_untied_default_passage_encoder():
    q = _Probe()
    r = R.UntiedDualEncoderRetriever(types.SimpleNamespace(query_side_retriever_training=False), q)
    assert r.passage_contriever is q                     # a bare encoder is shared ...
    wrapped = _Probe()
    wrapped.module = nn.Identity()                       # ... a wrapped one (DDP's `.module`) is copied (retrievers.py:116-118)
    r2 = R.UntiedDualEncoderRetriever(r.opt, wrapped)
    assert r2.passage_contriever is not wrapped and isinstance(r2.passage_contriever, _Probe)


def _reference_shaped_state_dict(prefixes, cfg):
    """the keys (and shapes) a reference Atlas checkpoint holds for the retriever: HF BertModel names under each prefix, incl. the
    persistent `embeddings.position_ids` buffer of modeling_bert.py:205"""
    H, I = cfg.hidden_size, cfg.intermediate_size
    shapes = {"embeddings.word_embeddings.weight": (cfg.vocab_size, H), "embeddings.position_embeddings.weight": (cfg.max_position_embeddings, H),
              "embeddings.token_type_embeddings.weight": (cfg.type_vocab_size, H), "embeddings.LayerNorm.weight": (H,),
              "embeddings.LayerNorm.bias": (H,), "embeddings.position_ids": (1, cfg.max_position_embeddings)}
    for l in range(cfg.num_hidden_layers):
        b = f"encoder.layer.{l}."
        for name, (o, i) in {"attention.self.query": (H, H), "attention.self.key": (H, H), "attention.self.value": (H, H),
                             "attention.output.dense": (H, H), "intermediate.dense": (I, H), "output.dense": (H, I)}.items():
            shapes[b + name + ".weight"], shapes[b + name + ".bias"] = (o, i), (o,)
        for name in ("attention.output.LayerNorm", "output.LayerNorm"):
            shapes[b + name + ".weight"], shapes[b + name + ".bias"] = (H,), (H,)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for pre in prefixes:
        for k, shp in shapes.items():
            sd[pre + k] = (torch.arange(shp[1]).expand(shp) if k.endswith("position_ids") else torch.randn(shp, generator=g) * 0.02)
    return sd


def test_reference_checkpoints_load_strict_in_both_wrappers():
    """model_io.py:122 does `model.load_state_dict(model_dict)` (strict); the keys must match exactly, in both directions"""
    cfg = R.BertConfigLite(vocab_size=40, num_hidden_layers=2, max_position_embeddings=24)
    opt = types.SimpleNamespace(query_side_retriever_training=False)
    tied = R.DualEncoderRetriever(opt, R.Contriever(cfg))
    sd = _reference_shaped_state_dict(["contriever."], cfg)
    assert tied.load_state_dict(sd, strict=True).missing_keys == []
    assert set(tied.state_dict()) == set(sd)
    untied = R.UntiedDualEncoderRetriever(opt, R.Contriever(cfg), R.Contriever(copy.deepcopy(cfg)))
    sd2 = _reference_shaped_state_dict(["query_contriever.", "passage_contriever."], cfg)
    untied.load_state_dict(sd2, strict=True)
    assert set(untied.state_dict()) == set(sd2)
    assert torch.equal(untied.passage_contriever.encoder.layer[1].output.dense.weight, sd2["passage_contriever.encoder.layer.1.output.dense.weight"])
    # the inference copy of atlas.py:59 keeps the integer buffer an integer buffer
    half = copy.deepcopy(tied).half().eval()
    assert half.contriever.embeddings.position_ids.dtype == torch.int64
    assert half.contriever.embeddings.word_embeddings.weight.dtype == torch.float16
