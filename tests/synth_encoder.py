"""Integer-deterministic BERT weights and token batches for the encoder golden fixtures (same splitmix64 / Irwin-Hall
generator as tests/synth.py: identical bits on any machine, no dependence on a torch RNG)."""
import hashlib

import numpy as np
import torch

import synth

# hidden 768 / 12 heads / 3072 is what the HIP encoder supports (= BERT-base); vocab is cut to keep generation fast
CASES = [
    {"name": "l2_ragged", "layers": 2, "vocab": 3000, "n": 6, "L": 40, "seed": 11, "holes": False},
    {"name": "l12_ragged", "layers": 12, "vocab": 3000, "n": 4, "L": 64, "seed": 12, "holes": False},
    {"name": "l2_holes_pad", "layers": 2, "vocab": 3000, "n": 5, "L": 96, "seed": 13, "holes": True},
]


def config_dict(case):
    return dict(vocab_size=case["vocab"], hidden_size=768, num_hidden_layers=case["layers"], num_attention_heads=12,
                intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0)


def _normal(shape, seed, scale, shift=0.0):
    n = int(np.prod(shape))
    x = synth.normal_f32(1, n, seed, 1.0)[0]   # ~N(0,1), float32, integer-deterministic
    return torch.from_numpy((x * np.float32(scale) + np.float32(shift)).reshape(shape).copy())


def _uniform_u32(n, seed):
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0xD1342543DE82EF95)) & synth._M
    return (synth._splitmix64(idx) >> np.uint64(32)).astype(np.int64)


def state_dict(case):
    """HF BertModel parameter names (what model_io.py:62-71 loads into retriever.contriever.*)"""
    s = case["seed"] * 1000
    c = config_dict(case)
    H, I = c["hidden_size"], c["intermediate_size"]
    sd = {}
    k = [0]

    def nxt():
        k[0] += 1
        return s + k[0]

    sd["embeddings.word_embeddings.weight"] = _normal((c["vocab_size"], H), nxt(), 0.05)
    sd["embeddings.word_embeddings.weight"][0].zero_()
    sd["embeddings.position_embeddings.weight"] = _normal((c["max_position_embeddings"], H), nxt(), 0.05)
    sd["embeddings.token_type_embeddings.weight"] = _normal((c["type_vocab_size"], H), nxt(), 0.05)
    sd["embeddings.LayerNorm.weight"] = _normal((H,), nxt(), 0.2, 1.0)
    sd["embeddings.LayerNorm.bias"] = _normal((H,), nxt(), 0.05)
    sd["embeddings.position_ids"] = torch.arange(c["max_position_embeddings"]).expand((1, -1))     # modeling_bert.py:205: in every checkpoint
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        for name, shape in (("attention.self.query", (H, H)), ("attention.self.key", (H, H)), ("attention.self.value", (H, H)),
                            ("attention.output.dense", (H, H)), ("intermediate.dense", (I, H)), ("output.dense", (H, I))):
            sd[p + name + ".weight"] = _normal(shape, nxt(), 0.04)
            sd[p + name + ".bias"] = _normal((shape[0],), nxt(), 0.03)
        for name in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + name + ".weight"] = _normal((H,), nxt(), 0.2, 1.0)
            sd[p + name + ".bias"] = _normal((H,), nxt(), 0.05)
    return sd


def state_sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        if k.endswith("position_ids"):       # the integer buffer is not a weight (and the fixtures' sha predates its presence here)
            continue
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    return h.hexdigest()


def inputs(case):
    n, L = case["n"], case["L"]
    u = _uniform_u32(n * L + 2 * n, case["seed"] * 77 + 5)
    ids = (u[: n * L] % (case["vocab"] - 1000) + 1000).astype(np.int64).reshape(n, L)
    lens = (u[n * L: n * L + n] % (L - L // 3) + L // 3 + 1).astype(np.int64)
    lens[0] = L
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    if case["holes"]:
        holes = (_uniform_u32(n * L, case["seed"] * 91 + 3).reshape(n, L) % 5) == 0
        mask = mask * (~holes)
        mask[:, 0] = 1
        mask[1, L // 4:] = 0                      # a short row: query-like padding
    ids = ids * mask
    ids[:, 0] = 101
    return torch.from_numpy(ids), torch.from_numpy(mask)
