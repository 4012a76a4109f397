"""Pre-tokenised passages of one shard, kept in pinned host memory, handed out in length-bucketed batches.

`Atlas.build_index` (src/atlas.py:61-88) formats and tokenises every passage of the shard again on every index refresh:
string formatting plus the HF tokenizer on ~4M passages per rank, on the host, while the GPU waits. The passages do not change
between refreshes -- only the retriever's weights do -- so the token ids are computed ONCE:

    store = TokenStore.from_passages(passages, tokenizer, opt.retriever_format, opt.text_maxlength, gpu_embedder_batch_size)

with exactly the tokenizer call of atlas.py:68-75 (`padding="longest"`, `truncation=True` and the reference's own
`max_length = min(text_maxlength, gpu_embedder_batch_size)` -- sic: the BATCH SIZE bounds the token count there; kept, because
the embeddings must be the ones the unchanged loop produces). Padding is not stored: a passage is its `length` real tokens.

Batches are formed by LENGTH, not by position: rows are ordered by token count and cut into groups of `batch_size`, so a batch is
as wide as its longest member instead of as the longest passage of 512 neighbours (H2D bytes and launch grids follow n x L; the
encoder itself only ever computes real tokens). Every batch carries the slab rows its passages belong to; the encoder's pooling
epilogue writes each embedding to its own row (`atlas_contriever_embed_rows`), so the slab comes out in passage order, bit-identical
to the position-ordered loop (the HIP encoder's result for a passage does not depend on what else is in its batch).
"""
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


class TokenStore:
    def __init__(self, tokens: torch.Tensor, offsets: np.ndarray, max_length: int):
        """tokens: int32 [T] (all passages back to back, in passage order); offsets: int64 [N + 1]"""
        assert tokens.dtype == torch.int32 and tokens.dim() == 1 and offsets.dtype == np.int64 and int(offsets[-1]) == tokens.numel()
        self.tokens = tokens.pin_memory() if torch.cuda.is_available() and not tokens.is_pinned() else tokens
        self._tok_np = self.tokens.numpy()
        self.offsets = offsets
        self.lengths = np.diff(offsets).astype(np.int64)
        self.max_length = int(max_length)

    def __len__(self) -> int:
        return int(self.lengths.shape[0])

    @property
    def n_tokens(self) -> int:
        return int(self.offsets[-1])

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_token_lists(cls, token_lists: Sequence[Sequence[int]], max_length: Optional[int] = None) -> "TokenStore":
        lens = np.fromiter((len(t) for t in token_lists), dtype=np.int64, count=len(token_lists))
        off = np.zeros(len(token_lists) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        flat = np.empty(int(off[-1]), dtype=np.int32)
        for i, t in enumerate(token_lists):
            flat[off[i]: off[i + 1]] = t
        return cls(torch.from_numpy(flat), off, int(max_length if max_length is not None else (lens.max() if len(lens) else 0)))

    @classmethod
    def from_passages(cls, passages, tokenizer, retriever_format: str, text_maxlength: int, gpu_embedder_batch_size: int,
                      chunk: int = 4096) -> "TokenStore":
        """Tokenise `passages` (dicts with the keys `retriever_format` names) once, with the call of src/atlas.py:66-75."""
        max_length = min(text_maxlength, gpu_embedder_batch_size)           # sic (atlas.py:74)
        pieces: List[np.ndarray] = []
        lens: List[np.ndarray] = []
        for a in range(0, len(passages), chunk):
            texts = [retriever_format.format(**example) for example in passages[a: a + chunk]]
            enc = tokenizer(texts, padding="longest", return_tensors="pt", max_length=max_length, truncation=True)
            ids, mask = enc["input_ids"], enc["attention_mask"].bool()
            n_real = mask.sum(dim=1)
            # the store keeps a length per passage: the mask must be the usual prefix mask, and single-segment inputs have type 0
            assert bool((mask == (torch.arange(mask.shape[1])[None, :] < n_real[:, None])).all()), "attention mask is not a prefix mask"
            if "token_type_ids" in enc:
                assert not bool(enc["token_type_ids"][mask].any()), "non-zero token_type_ids: not a single-segment passage encoding"
            pieces.append(ids[mask].to(torch.int32).numpy())
            lens.append(n_real.numpy().astype(np.int64))
        lens_all = np.concatenate(lens) if lens else np.zeros(0, np.int64)
        off = np.zeros(lens_all.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens_all, out=off[1:])
        flat = np.concatenate(pieces) if pieces else np.zeros(0, np.int32)
        return cls(torch.from_numpy(np.ascontiguousarray(flat, dtype=np.int32)), off, max_length)

    # ------------------------------------------------------------------ batches
    def plan(self, batch_size: int, bucket: bool = True, token_budget: Optional[int] = None) -> List[np.ndarray]:
        """the row groups of one refresh: by token count (stable, so equal lengths stay in passage order) or by position.
        token_budget: groups are cut by TOKENS instead of passages -- each takes passages (in that order) while their real tokens fit the
        budget, at most 2 * batch_size of them. The encoder works on packed tokens in 256-token tiles dealt to 8 x 32 workgroups: a batch
        of 65 536 tokens is 3 / 6 / 12 full rounds of tiles for its GEMMs, a batch of 512 passages x 132 tokens one round more with a
        few tiles in it (atlas_amd.refresh.TOKEN_BUDGET). Which passages share a batch does not change any embedding."""
        order = np.argsort(self.lengths, kind="stable") if bucket else np.arange(len(self), dtype=np.int64)
        if not token_budget:
            return [order[a: a + batch_size] for a in range(0, len(self), batch_size)]
        cum = np.zeros(len(self) + 1, dtype=np.int64)
        np.cumsum(self.lengths[order], out=cum[1:])
        groups, a, cap = [], 0, 2 * batch_size
        while a < len(self):
            b = int(np.searchsorted(cum, cum[a] + token_budget, side="right")) - 1      # most passages whose tokens fit
            b = min(max(b, a + 1), a + cap, len(self))
            groups.append(order[a:b])
            a = b
        return groups

    def fill(self, rows: np.ndarray, ids_out: torch.Tensor, mask_out: torch.Tensor) -> int:
        """write the [n, L] int64 `input_ids` / `attention_mask` of the passages `rows` into the (pinned) staging tensors;
        L = longest passage of the group. Returns L. Vectorised: no per-passage python."""
        lens = self.lengths[rows]
        n, L = int(rows.shape[0]), int(lens.max()) if rows.shape[0] else 0
        ids = ids_out.view(-1)[: n * L].view(n, L).numpy()
        mask = mask_out.view(-1)[: n * L].view(n, L).numpy()
        ids[...] = 0
        first = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=first[1:])
        which = np.repeat(np.arange(n, dtype=np.int64), lens)                 # passage of every token of the group
        pos = np.arange(int(first[-1]), dtype=np.int64) - first[which]        # its position inside the passage
        ids[which, pos] = self._tok_np[self.offsets[rows][which] + pos]
        mask[...] = np.arange(L, dtype=np.int64)[None, :] < lens[:, None]
        return L

    def batches(self, batch_size: int, bucket: bool = True) -> Iterator[Tuple[np.ndarray, torch.Tensor, torch.Tensor]]:
        """(rows, input_ids [n, L], attention_mask [n, L]) host tensors, for callers without their own staging"""
        for rows in self.plan(batch_size, bucket):
            n, L = rows.shape[0], int(self.lengths[rows].max())
            ids = torch.empty((n, L), dtype=torch.int64)
            mask = torch.empty((n, L), dtype=torch.int64)
            self.fill(rows, ids, mask)
            yield rows, ids, mask
