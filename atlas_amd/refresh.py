"""Streamed index refresh: the loop of `Atlas.build_index` (src/atlas.py:61-88) as a two-stream pipeline.

The reference formats + tokenises a batch on the host, moves it to the GPU, runs the fp16 retriever copy and scatters the
embeddings into `index.embeddings[:, a:b]`, one batch after the other, on every refresh. Here

  * the token ids come from a `TokenStore` built once (atlas_amd/token_store.py: same tokenizer call incl. the sic max_length,
    pinned host memory, length-bucketed batches);
  * batches are staged through pinned buffers and copied on a COPY stream while the encoder works on the previous batch on the
    compute stream; the host only blocks when all staging slots are in flight, nothing is synchronised per batch;
  * the pooled rows are written straight into the slab rows they belong to (`Contriever.embed_into(..., out_rows=)`);
  * the fp16 inference copy is refreshed IN PLACE from the training weights (`HalfMirror`) instead of
    `copy.deepcopy(retriever).half()` allocating 110M parameters anew for every refresh (atlas.py:54-59).

`build_index_streamed` has the signature of `Atlas.build_index` and can be bound in its place.
"""
import copy
import time
from typing import Iterable, Optional, Tuple

import torch

from . import dist_utils
from .token_store import TokenStore


class HalfMirror:
    """A persistent `.half().eval()` copy of a retriever whose parameters are re-cast in place before each refresh."""

    def __init__(self, retriever: torch.nn.Module):
        src = retriever.module if hasattr(retriever, "module") else retriever          # DDP / ShardedDDP (atlas.py:55-58)
        self.copy = copy.deepcopy(src).half().eval().requires_grad_(False)

    @torch.no_grad()
    def sync(self, retriever: torch.nn.Module) -> torch.nn.Module:
        src = retriever.module if hasattr(retriever, "module") else retriever
        for (name_d, dst), (name_s, s) in zip(self.copy.named_parameters(), src.named_parameters()):
            assert name_d == name_s and dst.shape == s.shape, (name_d, name_s)
            dst.copy_(s)                                                                  # fp32 -> fp16 on the device, RNE = .half()
        return self.copy


# Tokens of one streamed refresh batch: 256 token tiles of 256 = 32 per XCD, whole rounds of tiles for the 32 workgroups of an XCD in every
# GEMM of the encoder (csrc/encoder.hip: gemm_pt_kernel). Batches formed by passage COUNT (512 x ~132 tokens = 264 tiles) run one round
# more, nearly empty: 32.8k -> 34.4k passages/s, same process, slab bit-identical (profiles/r03/streamed_token_budget.txt)
TOKEN_BUDGET = 65536
# Round 6: a streamed refresh forms groups of BUDGET_SCALE x TOKEN_BUDGET tokens (which passages share a batch changes no embedding, and the
# batches here are ours, not the caller's 512): 512 / 1024 / 2048 / 4096 passages x 128 tokens per launch sequence run at 12.82 / 12.80 /
# 12.63 / 12.61 ms per 512 passages in one process (profiles/r06/enc_batch_size.txt) -- every launch's ramp, first fetch and last-tile tail
# spread over four times the tiles. Costs workspace only (~4 GB of the 288).
BUDGET_SCALE = 4


class IndexRefresher:
    def __init__(self, index, contriever_fp16, max_batch: int, max_len: int, depth: int = 3):
        """index: HipDistributedIndex with its slab allocated (init_embeddings); contriever_fp16: atlas_amd.retrievers.Contriever
        in fp16 on the slab's device (the `.half().eval()` copy of atlas.py:59)."""
        self.index, self.enc = index, contriever_fp16
        self.dev = index._slab.device
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.depth, self.max_batch, self.max_len = depth, max_batch, max_len
        # (staging holds BUDGET_SCALE x max_batch x max_len token slots; a batch cut by tokens may have up to 2 x BUDGET_SCALE x max_batch shorter
        #  passages in them)
        self.stage_batch = max_batch * BUDGET_SCALE
        mk = lambda: torch.empty((self.stage_batch, max_len), dtype=torch.int64).pin_memory()     # noqa: E731
        self._pin = [(mk(), mk(), torch.empty(2 * self.stage_batch, dtype=torch.int64).pin_memory()) for _ in range(depth)]
        self._dev = [(torch.empty((self.stage_batch, max_len), dtype=torch.int64, device=self.dev),
                      torch.empty((self.stage_batch, max_len), dtype=torch.int64, device=self.dev),
                      torch.empty(2 * self.stage_batch, dtype=torch.int64, device=self.dev)) for _ in range(depth)]
        self._ready = [torch.cuda.Event() for _ in range(depth)]      # H2D of the slot finished
        self._free = [torch.cuda.Event() for _ in range(depth)]       # encoder finished reading the slot
        self._used = [False] * depth
        self._turn = 0
        # host-side seconds of the refreshes run so far: filling the pinned staging buffers from the token store, and blocked on a staging
        # slot the encoder still reads (= the device is the bottleneck); bench.py's full-shard leg reports their share
        self.host_seconds = {"fill": 0.0, "slot_wait": 0.0, "launch": 0.0}

    def _slot(self) -> int:
        s = self._turn % self.depth
        self._turn += 1
        if self._used[s]:
            t = time.perf_counter()
            self._free[s].synchronize()                                # only when `depth` batches are already in flight
            self.host_seconds["slot_wait"] += time.perf_counter() - t
        return s

    def _launch(self, s: int, n: int, L: int, rows: Optional[bool], row_offset: int = 0) -> None:
        """H2D of slot s on the copy stream, then the encoder on the compute stream, rows written into the slab"""
        compute = torch.cuda.current_stream(self.dev)
        pi, pm, pr = self._pin[s]
        di, dm, dr = self._dev[s]
        with torch.cuda.stream(self.copy_stream):
            ids_d = di.view(-1)[: n * L].view(n, L)                    # contiguous [n, L] views of the staging buffers
            mask_d = dm.view(-1)[: n * L].view(n, L)
            ids_d.copy_(pi.view(-1)[: n * L].view(n, L), non_blocking=True)
            mask_d.copy_(pm.view(-1)[: n * L].view(n, L), non_blocking=True)
            if rows:
                dr[:n].copy_(pr[:n], non_blocking=True)
            self._ready[s].record(self.copy_stream)
        compute.wait_event(self._ready[s])
        if rows:
            self.enc.embed_into(self.index._slab, ids_d, mask_d, out_rows=dr[:n])
        else:
            self.enc.embed_into(self.index._slab[row_offset: row_offset + n], ids_d, mask_d)
        self._free[s].record(compute)
        self._used[s] = True

    @torch.no_grad()
    def run(self, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], row_offset: int = 0) -> int:
        """batches: (input_ids, attention_mask) host tensors [n, L] (n <= max_batch, L <= max_len), in slab row order
        starting at row_offset. Returns the number of rows written. Asynchronous: call torch.cuda.synchronize() (or use
        the slab on the current stream) afterwards."""
        row = row_offset
        for ids, mask in batches:
            s = self._slot()
            n, L = ids.shape
            pi, pm, _ = self._pin[s]
            pi.view(-1)[: n * L].view(n, L).copy_(ids)
            pm.view(-1)[: n * L].view(n, L).copy_(mask)
            self._launch(s, n, L, rows=False, row_offset=row)
            row += n
        self.index._pmax = None                                        # row norms changed: re-certify on the next search
        return row - row_offset

    def plan(self, store: TokenStore, batch_size: Optional[int] = None, bucket: bool = True, token_budget: Optional[int] = None):
        """the row groups `run_store` launches for these arguments"""
        batch_size = batch_size or self.max_batch
        assert batch_size <= self.max_batch and store.max_length <= self.max_len and len(store) == self.index._slab.shape[0]
        group_batch = batch_size
        if token_budget is None and bucket and len(store) and batch_size * store.n_tokens / len(store) >= 0.75 * TOKEN_BUDGET:
            token_budget, group_batch = TOKEN_BUDGET * BUDGET_SCALE, batch_size * BUDGET_SCALE
        plan = store.plan(group_batch, bucket, token_budget)
        if token_budget:                                               # every group has to fit the staging buffers as [n, L]
            slots = self.stage_batch * self.max_len
            out = []
            for grp in plan:
                parts = [grp]
                while any(g.shape[0] > 1 and g.shape[0] * int(store.lengths[g].max()) > slots for g in parts) or any(g.shape[0] > 2 * self.stage_batch for g in parts):
                    parts = [h for g in parts for h in (g[: g.shape[0] // 2], g[g.shape[0] // 2:])]
                out += parts
            plan = out
        return plan

    @torch.no_grad()
    def run_store(self, store: TokenStore, batch_size: Optional[int] = None, bucket: bool = True, repeat: int = 1,
                  token_budget: Optional[int] = None) -> int:
        """One refresh of the whole shard from a token store (`repeat` > 1: back-to-back refreshes, for sustained-rate timing).
        Row r of the slab receives the embedding of passage r of the store. Asynchronous like `run`.
        token_budget: batches are cut by tokens (TokenStore.plan); default: TOKEN_BUDGET when the batches are formed by length and
        batch_size passages of the store's mean length reach it, else by passage count."""
        plan = self.plan(store, batch_size, bucket, token_budget)
        for _ in range(repeat):
            for rows in plan:
                s = self._slot()
                pi, pm, pr = self._pin[s]
                t0 = time.perf_counter()
                L = store.fill(rows, pi, pm)
                pr[: rows.shape[0]].copy_(torch.from_numpy(rows))
                t1 = time.perf_counter()
                self._launch(s, int(rows.shape[0]), L, rows=True)
                self.host_seconds["fill"] += t1 - t0
                self.host_seconds["launch"] += time.perf_counter() - t1
        self.index._pmax = None
        return len(store) * repeat


def _passages_fingerprint(passages) -> tuple:
    """a content fingerprint of a passage list: (hash of id, title and text of ~64 spread entries, total length of ALL texts).
    The sampled hash is O(1); the total text length is one C-level pass over the list (`sum(map(len, ...))`: ~0.25 s per 4M passages, against
    the ~100 s their refresh takes) and notices an in-place edit of ANY passage that changes its length (round 5 had dropped the
    full pass, so an edited unsampled passage kept its stale tokens). Still a heuristic for same-length edits of unsampled passages: a caller
    that edits passages in place calls `invalidate_refresh_state(index)`; a NEW list object (what `index_io.load_passages` returns) is always
    re-tokenised."""
    import operator

    n = len(passages)
    if n == 0:
        return (0, 0)
    picks = sorted({0, n - 1, *range(0, n, max(1, n // 62))})
    sampled = hash(tuple((passages[i].get("id"), passages[i].get("title"), passages[i].get("text")) for i in picks))
    try:
        total = sum(map(len, map(operator.itemgetter("text"), passages)))
    except (KeyError, TypeError):                                    # (a passage without a text, or a text that is not a string)
        total = sum(len(p.get("text") or "") for p in passages)
    return (sampled, total)


def invalidate_refresh_state(index) -> None:
    """forget the token store / fp16 mirror `build_index_streamed` keeps on `index`: the next refresh tokenises the passages again"""
    index.__dict__.get("_refresh_state", {}).clear()


@torch.no_grad()
def build_index_streamed(self, index, passages, gpu_embedder_batch_size, logger=None):
    """`Atlas.build_index` (src/atlas.py:61-88) with the streamed refresh underneath; `self` is the Atlas module
    (`Atlas.build_index = atlas_amd.refresh.build_index_streamed`, or `types.MethodType(build_index_streamed, model)`).

    First call: tokenises `passages` once into a TokenStore and builds the persistent fp16 mirror of the retriever; every later
    call re-casts the weights in place and streams the stored tokens. Same slab as the reference's loop run on the same encoder."""
    state = index.__dict__.setdefault("_refresh_state", {})
    # the token store is reused while `passages` is the same list with the same content at its ends and in its middle (a list edited
    # in place keeps its id and, often, its length: the fingerprint is what notices)
    key = (len(passages), _passages_fingerprint(passages))
    if state.get("passages_ref") is not passages or state.get("passages_key") != key:      # (`is` on a held reference: a freed list's id() may be reused)
        state.clear()
        state["passages_ref"] = passages
        state["store"] = TokenStore.from_passages(passages, self.retriever_tokenizer, self.opt.retriever_format, self.opt.text_maxlength,
                                                  gpu_embedder_batch_size)
        state["passages_key"] = key
    if "mirror" not in state:
        state["mirror"] = HalfMirror(self.retriever)
    half = state["mirror"].sync(self.retriever)
    encoder = getattr(half, half.passage_role) if hasattr(half, "passage_role") else half
    store = state["store"]
    if len(store):
        if state.get("refresher_key") != (id(encoder), gpu_embedder_batch_size):
            state["refresher"] = IndexRefresher(index, encoder, gpu_embedder_batch_size, max(store.max_length, 1))
            state["refresher_key"] = (id(encoder), gpu_embedder_batch_size)
        index._check_slab()
        state["refresher"].index = index
        total = state["refresher"].run_store(store, gpu_embedder_batch_size)
        torch.cuda.current_stream(index._slab.device).synchronize()
    else:
        total = 0
    dist_utils.barrier()
    if logger is not None:
        logger.info(f"{total} passages encoded on process: {dist_utils.get_rank()}")
    if not index.is_index_trained():
        index.train_index()
