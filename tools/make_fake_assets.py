"""A stand-in asset directory for DRY RUNS of scripts/real_assets.sh (no real checkpoint, vocabulary or corpus exists offline): a random-init
BERT-base-shaped Contriever in the HF layout (config.json + model.safetensors), a WordPiece vocab.txt of made-up words, a passages jsonl and a
queries jsonl whose answers occur in the corpus. Everything is labelled fake; numbers measured on it say the pipeline runs, nothing else.
    python tools/make_fake_assets.py OUT_DIR [--layers 12] [--passages 20000]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--passages", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=256)
    args = ap.parse_args()
    from safetensors.torch import save_file

    from atlas_amd import retrievers as R

    ck = os.path.join(args.out, "contriever")
    os.makedirs(ck, exist_ok=True)
    torch.manual_seed(3)
    m = R.Contriever(R.BertConfigLite(num_hidden_layers=args.layers))
    c = m.config
    json.dump({"model_type": "bert", "architectures": ["Contriever"], "vocab_size": c.vocab_size, "hidden_size": c.hidden_size,
               "num_hidden_layers": c.num_hidden_layers, "num_attention_heads": c.num_attention_heads, "intermediate_size": c.intermediate_size,
               "max_position_embeddings": c.max_position_embeddings, "type_vocab_size": c.type_vocab_size, "layer_norm_eps": c.layer_norm_eps,
               "hidden_act": "gelu", "_fake": "random-init weights (tools/make_fake_assets.py)"}, open(os.path.join(ck, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in m.state_dict().items() if "position_ids" not in k}, os.path.join(ck, "model.safetensors"))
    rng = np.random.default_rng(11)
    syll = ["ka", "lo", "mi", "ne", "su", "ta", "ri", "vo", "pe", "du", "sha", "gra", "tol", "ben", "kir", "zum"]
    words = sorted({"".join(rng.choice(syll, size=int(rng.integers(1, 4)))) for _ in range(6000)})
    special = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab = special + list("abcdefghijklmnopqrstuvwxyz0123456789") + ["##" + ch for ch in "abcdefghijklmnopqrstuvwxyz0123456789"] + words
    vocab += [f"[filler{i}]" for i in range(c.vocab_size - len(vocab))]
    assert len(vocab) == c.vocab_size and vocab[101] == "[CLS]" and vocab[102] == "[SEP]"
    open(os.path.join(ck, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
    json.dump({"do_lower_case": True, "tokenizer_class": "BertTokenizer", "model_max_length": 512}, open(os.path.join(ck, "tokenizer_config.json"), "w"))
    with open(os.path.join(args.out, "passages.jsonl"), "w") as f:
        for i in range(args.passages):
            f.write(json.dumps({"id": str(i), "title": " ".join(rng.choice(words, size=2)), "section": "", "text": " ".join(rng.choice(words, size=int(rng.integers(20, 140))))}) + "\n")
    with open(os.path.join(args.out, "queries.jsonl"), "w") as f:
        lines = open(os.path.join(args.out, "passages.jsonl")).read().splitlines()
        for j in range(args.queries):
            p = json.loads(lines[(j * 7919) % len(lines)])
            toks = p["text"].split()
            f.write(json.dumps({"question": " ".join(toks[:8]), "answers": [" ".join(toks[8:10])]}) + "\n")
    print(args.out)


if __name__ == "__main__":
    main()
