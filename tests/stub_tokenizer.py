"""A stand-in for the HF BERT tokenizer (no vocab files offline): call-compatible with the calls of src/atlas.py:68-75, 152-158,
185-191 -- words -> ids by a stable hash, [CLS] .. [SEP] framing, padding 'longest' / 'max_length', truncation to max_length,
'pt' tensors incl. token_type_ids (BERT tokenizers return them, and atlas.py:78 passes `**batch_enc` on)."""
import torch


class HashTokenizer:
    def __init__(self, vocab_size=1000):
        self.vocab_size = vocab_size
        self.calls = []

    def encode_one(self, text, max_length):
        ids = [101] + [103 + (sum(map(ord, w)) * 31 + len(w)) % (self.vocab_size - 200) for w in text.split()]
        return ids[: max_length - 1] + [102]

    def __call__(self, batch, padding=None, return_tensors=None, max_length=None, truncation=None):
        assert return_tensors == "pt" and truncation is True
        self.calls.append(dict(n=len(batch), padding=padding, max_length=max_length))
        rows = [self.encode_one(text, max_length) for text in batch]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        input_ids = torch.zeros((len(rows), width), dtype=torch.int64)
        mask = torch.zeros((len(rows), width), dtype=torch.int64)
        for i, r in enumerate(rows):
            input_ids[i, : len(r)] = torch.tensor(r)
            mask[i, : len(r)] = 1
        return {"input_ids": input_ids, "token_type_ids": torch.zeros_like(input_ids), "attention_mask": mask}
