"""The TRAINING forward of the Contriever module: the one call of the reference that needs autograd.

`Atlas.forward` embeds the query (and, unless `query_side_retriever_training`, the retrieved passages) with the retriever in
train mode, under autograd, and back-propagates the retriever loss through it (src/atlas.py:452-465); `set_dropout` has set every
`nn.Dropout` of the model to `opt.dropout` by then (src/model_io.py:103, src/util.py:161-164). That step is OUTSIDE the path this
package accelerates (index refresh, query embedding and search run under `torch.no_grad()` on the HIP encoder and never come
here) -- but a drop-in retriever has to survive it, so `Contriever.forward` hands exactly two cases to this file:

    * autograd is on and a parameter requires grad                       (retriever training)
    * the module is in train mode with a non-zero dropout probability    (the reference would drop activations; the HIP
                                                                          encoder is an eval-mode encoder)

Everything here is written with torch operators on the module's own parameters, so torch's autograd provides the backward and the
optimizer / DDP / ShardedDDP wrappers of the reference see ordinary leaves. The operations, their order and the dtype at every step
are those of the reference's BERT (src/modeling_bert.py: embeddings 213-247, LayerNorm 104-114, self-attention 290-366 with the
softmax in fp32 :352, self-output 382-387, intermediate 448-451, output 461-466, encoder loop incl. gradient checkpointing
575-625) and of `Contriever.forward` (src/retrievers.py:49-60), including the four dropout sites (:246, :356, :384, :463) in that
order -- on the CPU, with the same generator state, the result and the gradients are bit-identical to the reference module
(tests/test_retriever_training.py). It is plumbing around the product, not the product: no HIP kernel is involved and nothing in
`bench.py` or the inference path may end up here (`Contriever.last_path` says which way a call went; the tests assert it).
"""
import math

import torch
import torch.nn.functional as F
import torch.utils.checkpoint


def _layer_norm(ln, x):
    """BertLayerNorm as the reference computes it (modeling_bert.py:104-114) -- NOT torch's LayerNorm: the scale is the root of the
    UNCENTRED second moment, statistics in fp32, the normalised value cast to a 16-bit weight's dtype before the affine step"""
    x32 = x.float()
    centred = x - torch.mean(x32, dim=-1, keepdim=True)
    y = centred * torch.rsqrt(torch.mean(x32.pow(2), dim=-1, keepdim=True) + ln.variance_epsilon)
    w = ln.weight
    if w.dtype.itemsize == 2:                      # fp16 / bf16 weights
        y = y.to(w.dtype)
    return w * y + ln.bias


def _embed(emb, input_ids, token_type_ids):
    seq = input_ids.shape[1]
    position_ids = emb.position_ids[:, :seq]
    if token_type_ids is None:
        token_type_ids = torch.zeros(input_ids.shape, dtype=torch.long, device=emb.position_ids.device)
    x = emb.word_embeddings(input_ids) + emb.token_type_embeddings(token_type_ids)
    x += emb.position_embeddings(position_ids)
    x = _layer_norm(emb.LayerNorm, x.float()).type_as(x)
    return emb.dropout(x)


def _heads(x, n_heads):
    b, s, h = x.shape
    return x.view(b, s, n_heads, h // n_heads).permute(0, 2, 1, 3)


def _layer(layer, n_heads, x, ext_mask):
    att = layer.attention.self
    q, k, v = _heads(att.query(x), n_heads), _heads(att.key(x), n_heads), _heads(att.value(x), n_heads)
    scores = torch.matmul(q, k.transpose(-1, -2))
    scores = scores / math.sqrt(q.shape[-1])
    scores = scores + ext_mask
    probs = F.softmax(scores.float(), dim=-1).type_as(scores)
    probs = att.dropout(probs)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
    ctx = ctx.view(ctx.shape[0], ctx.shape[1], -1)

    so = layer.attention.output                                   # BertSelfOutput
    a = so.dropout(so.dense(ctx)) + x
    a = _layer_norm(so.LayerNorm, a.float()).type_as(a)

    inter = F.gelu(layer.intermediate.dense(a))                   # ACT2FN["gelu"]: exact erf
    out = layer.output                                            # BertOutput
    y = out.dropout(out.dense(inter)) + a
    return _layer_norm(out.LayerNorm, y.float()).type_as(y)


def training_forward(model, input_ids, attention_mask, token_type_ids=None, normalize=False):
    """`Contriever.forward` (retrievers.py:22-60) of `model` (atlas_amd.retrievers.Contriever) with torch operators"""
    c = model.config
    dtype = model.embeddings.word_embeddings.weight.dtype
    if attention_mask is None:
        attention_mask = torch.ones(input_ids.shape, device=input_ids.device)
    # transformers 4.18 get_extended_attention_mask (the version the reference pins, requirements.txt:2)
    ext_mask = (1.0 - attention_mask[:, None, None, :].to(dtype)) * -10000.0
    x = _embed(model.embeddings, input_ids, token_type_ids)
    checkpointing = model.encoder.gradient_checkpointing and model.training
    for layer in model.encoder.layer:
        if checkpointing:
            x = torch.utils.checkpoint.checkpoint(_layer, layer, c.num_attention_heads, x, ext_mask, use_reentrant=False)
        else:
            x = _layer(layer, c.num_attention_heads, x, ext_mask)
    # retrievers.py:49-59: padded positions zeroed, then one of three poolings over the sequence axis
    if c.pooling not in _POOLINGS:
        raise ValueError(f"pooling={c.pooling!r}: the reference knows 'average', 'sqrt', 'cls' (retrievers.py:51-56)")
    hidden = x.masked_fill(attention_mask.unsqueeze(-1) == 0, 0.0)
    emb = _POOLINGS[c.pooling](hidden, attention_mask.sum(dim=1, keepdim=True))
    return F.normalize(emb, dim=-1) if normalize else emb


_POOLINGS = {
    "average": lambda hidden, count: hidden.sum(dim=1) / count,                       # integer count: true division in the sum's dtype
    "sqrt": lambda hidden, count: hidden.sum(dim=1) / count.float().sqrt(),           # fp32 divisor: promotes
    "cls": lambda hidden, count: hidden[:, 0],
}
