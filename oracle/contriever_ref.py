"""Torch restatement of the reference's Contriever encoder. TEST INFRASTRUCTURE ONLY.

PINNED: tests/test_encoder_golden.py checks this file against outputs of the reference's OWN module
(src/retrievers.py + src/modeling_bert.py run unmodified by tests/golden/make_golden_encoder.py, which supplies the
three transformers==4.18 helpers the installed transformers 5.x no longer has); on the generating machine the
restatement is bit-identical to the reference in fp32 and in fp16 (`.half()`), for 2- and 12-layer models, ragged
masks, masks with holes. The reference itself cannot travel to the GPU box (and needs those shims), so the GPU parity
tests use this restatement plus the committed fixtures. It follows the reference op by op, with the same tensor dtypes
at every step, so that running it on fp16 weights reproduces what `copy.deepcopy(retriever).half().eval()` computes in
`Atlas.build_index` (src/atlas.py:54-59, 78):

    BertEmbeddings.forward        modeling_bert.py:213-247   word + token_type (+= position), LayerNorm(x.float()).type_as
    BertLayerNorm.forward         modeling_bert.py:104-114   NON-standard: (x - mean) * rsqrt(mean(x^2) + eps), fp32 stats,
                                                             cast to the weight dtype, then weight * y + bias in that dtype
    BertSelfAttention.forward     modeling_bert.py:290-366   QK^T (-> dtype), / sqrt(64), + extended mask, softmax in fp32
                                                             .type_as, PV
    BertSelfOutput / BertOutput   modeling_bert.py:382-387, 461-466   dense, + residual, LayerNorm(x.float()).type_as
    BertIntermediate              modeling_bert.py:448-451   dense, exact-erf GELU (ACT2FN["gelu"])
    get_extended_attention_mask   transformers 4.18 modeling_utils: (1 - mask[:, None, None, :]).to(dtype) * -10000.0
    Contriever.forward            retrievers.py:49-60        masked_fill(~mask, 0), sum(dim=1) / mask.sum(dim=1)[..., None]

Parameter names match the HF BertModel state dict (`embeddings.word_embeddings.weight`,
`encoder.layer.{i}.attention.self.query.weight`, ...) so real Contriever checkpoints load into it unchanged.
BERT-base shape is implied by README.md:267-274 + retrievers.py:13 (the facebook/contriever config.json is not vendored).
"""
import math

import torch
import torch.nn as nn


class BertConfigLite:
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 initializer_range=0.02, pad_token_id=0):
        self.__dict__.update(locals())
        del self.__dict__["self"]


class BertLayerNorm(nn.Module):
    def __init__(self, hidden_size, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):                       # modeling_bert.py:104-114
        mean = hidden_states.to(torch.float32).mean(-1, keepdim=True)
        variance = hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hidden_states = (hidden_states - mean) * torch.rsqrt(variance + self.variance_epsilon)
        if self.weight.dtype in [torch.float16, torch.bfloat16]:
            hidden_states = hidden_states.to(self.weight.dtype)
        return self.weight * hidden_states + self.bias


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=c.pad_token_id)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = BertLayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(c.max_position_embeddings).expand((1, -1)))   # modeling_bert.py:205 (persistent)

    def forward(self, input_ids, token_type_ids=None):      # modeling_bert.py:213-247
        L = input_ids.shape[1]
        position_ids = torch.arange(L, device=input_ids.device)[None, :]
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        embeddings = self.word_embeddings(input_ids) + self.token_type_embeddings(token_type_ids)
        embeddings += self.position_embeddings(position_ids)
        return self.LayerNorm(embeddings.float()).type_as(embeddings)


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.h, self.dh = c.num_attention_heads, c.hidden_size // c.num_attention_heads
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)

    def _t(self, x):
        return x.view(x.shape[0], x.shape[1], self.h, self.dh).permute(0, 2, 1, 3)

    def forward(self, x, ext_mask):                          # modeling_bert.py:290-366
        q, k, v = self._t(self.query(x)), self._t(self.key(x)), self._t(self.value(x))
        scores = torch.matmul(q, k.transpose(-1, -2))
        scores = scores / math.sqrt(self.dh)
        scores = scores + ext_mask
        probs = nn.functional.softmax(scores.float(), dim=-1).type_as(scores)
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
        return ctx.view(ctx.shape[0], ctx.shape[1], self.h * self.dh)


class _SelfOutput(nn.Module):
    def __init__(self, c, in_features):
        super().__init__()
        self.dense = nn.Linear(in_features, c.hidden_size)
        self.LayerNorm = BertLayerNorm(c.hidden_size, eps=c.layer_norm_eps)

    def forward(self, hidden_states, input_tensor):          # modeling_bert.py:382-387 / 461-466
        hidden_states = self.dense(hidden_states)
        hidden_states = hidden_states + input_tensor
        return self.LayerNorm(hidden_states.float()).type_as(hidden_states)


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _SelfOutput(c, c.hidden_size)

    def forward(self, x, ext_mask):
        return self.output(self.self(x, ext_mask), x)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)

    def forward(self, x):                                    # modeling_bert.py:448-451, ACT2FN["gelu"] = exact erf
        return nn.functional.gelu(self.dense(x))


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _Attention(c)
        self.intermediate = _Intermediate(c)
        self.output = _SelfOutput(c, c.intermediate_size)

    def forward(self, x, ext_mask):
        a = self.attention(x, ext_mask)
        return self.output(self.intermediate(a), a)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])

    def forward(self, x, ext_mask):
        for layer in self.layer:
            x = layer(x, ext_mask)
        return x


class ContrieverRef(nn.Module):
    """retrievers.py:16-60 (pooling='average') on top of the BertModel forward, modeling_bert.py:918-1045."""

    def __init__(self, config=None, seed=99):
        super().__init__()
        self.config = config or BertConfigLite()
        self.embeddings = _Embeddings(self.config)
        self.encoder = _Encoder(self.config)
        g = torch.Generator().manual_seed(seed)
        for name, p in self.named_parameters():              # modeling_bert.py:747-761 (_init_weights)
            with torch.no_grad():
                if name.endswith("LayerNorm.weight"):
                    p.fill_(1.0)
                elif name.endswith("bias"):
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * self.config.initializer_range)
        with torch.no_grad():
            self.embeddings.word_embeddings.weight[self.config.pad_token_id].zero_()

    def randomize_affine(self, seed=7, scale=0.2):
        """non-trivial LayerNorm weights / all biases, so parity tests exercise every term"""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("bias"):
                    p.copy_(torch.randn(p.shape, generator=g) * scale * 0.1)
                elif name.endswith("LayerNorm.weight"):
                    p.copy_(1.0 + torch.randn(p.shape, generator=g) * scale)
        return self

    @torch.no_grad()
    def last_hidden(self, input_ids, attention_mask, token_type_ids=None):
        dtype = self.embeddings.word_embeddings.weight.dtype
        ext = (1.0 - attention_mask[:, None, None, :].to(dtype)) * -10000.0     # transformers 4.18 get_extended_attention_mask
        return self.encoder(self.embeddings(input_ids, token_type_ids), ext)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask, token_type_ids=None, normalize=False, pooling="average"):
        last_hidden = self.last_hidden(input_ids, attention_mask, token_type_ids)
        last_hidden = last_hidden.masked_fill(~attention_mask[..., None].bool(), 0.0).clone()   # retrievers.py:50
        if pooling == "average":
            emb = last_hidden.sum(dim=1).clone() / attention_mask.sum(dim=1)[..., None].clone()   # retrievers.py:52
        elif pooling == "sqrt":
            emb = last_hidden.sum(dim=1) / torch.sqrt(attention_mask.sum(dim=1)[..., None].float())   # retrievers.py:54
        elif pooling == "cls":
            emb = last_hidden[:, 0]                                                               # retrievers.py:56
        if normalize:
            emb = torch.nn.functional.normalize(emb, dim=-1).clone()
        return emb
