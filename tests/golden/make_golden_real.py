"""Golden encoder fixture from a REAL Contriever checkpoint, through the REFERENCE's own modules (src/retrievers.py over src/modeling_bert.py,
imported unmodified exactly as tests/golden/make_golden_encoder.py does). Build container only (needs /root/reference):

    python tests/golden/make_golden_real.py --checkpoint $ATLAS_CONTRIEVER_DIR [--passages corpus.jsonl] [--n 48] [--max-length 128]

The checkpoint directory is the HF layout `facebook/contriever` ships (config.json + model.safetensors | pytorch_model.bin [+ tokenizer files]);
neither it nor a tokenizer vocabulary exists offline, which is why every committed encoder fixture is on synthetic weights (VERDICT r05 missing #3).
With tokenizer files in the directory (or --tokenizer DIR) and --passages, the inputs are the first --n passages of the corpus tokenised by the
call of src/atlas.py:66-75; otherwise uniform random ids framed by [CLS] / [SEP] with ragged lengths (seeded).

Output: tests/golden/enc_real.npz = input_ids, attention_mask, the reference's embeddings of the fp32 model and of its `.half()` copy (CPU), a
sha256 of the checkpoint's tensors (so a test never compares against another checkpoint's numbers), torch version and CPU capability.
tests/test_gpu_encoder_real.py holds the HIP encoder to it on the GPU box (needs the same checkpoint there: ATLAS_CONTRIEVER_DIR)."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden_encoder as mge  # noqa: E402  (the reference import + the transformers 4.18 restatements)


def checkpoint_sha(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        if "position_ids" in k:
            continue
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().float().numpy().tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--tokenizer", default=None, help="directory with the tokenizer files (default: the checkpoint directory, if it has them)")
    ap.add_argument("--passages", default=None, help="jsonl with id / title / text (src/index_io.py:17-62)")
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--max-length", type=int, default=128)
    ap.add_argument("--out", default=os.path.join(HERE, "enc_real.npz"))
    args = ap.parse_args()
    torch.set_num_threads(1)
    from atlas_amd import retrievers

    mine = retrievers.Contriever.from_pretrained(args.checkpoint).eval()        # (only to read config + tensors in one place; nothing is computed with it)
    sd = {k: v for k, v in mine.state_dict().items()}
    ref = mge.import_reference_contriever()
    from transformers.models.bert.configuration_bert import BertConfig

    c = mine.config
    config = BertConfig(vocab_size=c.vocab_size, hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                        intermediate_size=c.intermediate_size, max_position_embeddings=c.max_position_embeddings, type_vocab_size=c.type_vocab_size,
                        layer_norm_eps=c.layer_norm_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = mge.bind_4_18(ref.Contriever(config))
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("position_ids" in k for k in res.missing_keys), res
    model.eval()

    tok_dir = args.tokenizer or args.checkpoint
    tokenizer = None
    if any(os.path.exists(os.path.join(tok_dir, f)) for f in ("vocab.txt", "tokenizer.json")):
        import transformers

        tokenizer = transformers.AutoTokenizer.from_pretrained(tok_dir)
    if tokenizer is not None and args.passages:
        from atlas_amd import index_io

        ps = [p for p in index_io.load_passages([args.passages], args.n) if p is not None][: args.n]
        enc = tokenizer(["{title} {text}".format(**p) for p in ps], padding="longest", return_tensors="pt", max_length=args.max_length, truncation=True)
        ids, mask, source = enc["input_ids"], enc["attention_mask"], f"first {len(ps)} passages of {os.path.basename(args.passages)}, real tokenizer"
    else:
        g = torch.Generator().manual_seed(20260930)
        lens = torch.randint(16, args.max_length + 1, (args.n,), generator=g)
        lens[0] = args.max_length
        ids = torch.randint(1000, c.vocab_size, (args.n, args.max_length), generator=g)
        mask = (torch.arange(args.max_length)[None, :] < lens[:, None]).long()
        ids = ids * mask
        ids[:, 0] = 101
        ids[torch.arange(args.n), lens - 1] = 102
        source = "uniform random token ids, ragged lengths (no tokenizer files / corpus given)"
    with torch.no_grad():
        e32 = model(input_ids=ids, attention_mask=mask).float().numpy()
        m16 = mge.bind_4_18(model.half())
        e16 = m16(input_ids=ids, attention_mask=mask).numpy()
    np.savez_compressed(args.out, input_ids=ids.numpy(), attention_mask=mask.numpy(), emb_fp32=e32, emb_fp16=e16,
                        checkpoint_sha=np.frombuffer(checkpoint_sha(sd).encode(), dtype=np.uint8),
                        meta=np.frombuffer(json.dumps({"source": source, "torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(),
                                                       "layers": c.num_hidden_layers, "pooling": c.pooling}).encode(), dtype=np.uint8))
    print(f"{args.out}: {ids.shape[0]} passages x {ids.shape[1]} tokens ({source}); max|e| fp32 {np.abs(e32).max():.3f}, "
          f"fp16 vs fp32 max|d| / max|e| = {np.abs(e16.astype(np.float32) - e32).max() / np.abs(e32).max():.2e}")


if __name__ == "__main__":
    main()
