"""HipDistributedIndex — drop-in for the reference's flat `DistributedIndex` (src/index.py:43-160).

Same public surface, same argument meaning, same error behaviour:

    index.init_embeddings(passages, dim=768)            index.py:48-53
    index.embeddings[:, a:b] = X.T                      atlas.py:79   (a (d,N) view of the slab)
    index.search_knn(queries, topk) -> (docs, scores)   index.py:122-157
    index.save_index / load_index                       index.py:61-111  (same files on disk)
    index.is_index_trained() / train_index()            index.py:159-160
    index.doc_map, index.is_in_gpu

What differs underneath:
  * the slab is (N, d) row-major fp16 in HBM (coalesced row reads); `embeddings` is its
    transposed view, so callers still see the reference's (d, N) tensor;
  * `_compute_scores_and_indices` is one fused HIP scan (no (B, N) score matrix), returning the
    canonical result: scores = correctly rounded fp16 of the exact inner product, ties broken by
    lowest passage id — the summation-order / tie-order independent member of the family of
    results the reference's backend may produce (include/atlas_hip.h);
  * the distributed search uses one packed (score,id) all-gather instead of 4*W pickled
    gathers, and ships passage text only for the k winners of each query, to the rank that asked.

There is no CPU path: every compute call goes through atlas_amd/_lib.py and raises if the HIP
library is missing or the slab is not on a GPU.
"""
import logging
import os
import pickle
from typing import List, Optional, Tuple

import warnings

import numpy as np
import torch

from . import _lib, dist_utils

EMBEDDINGS_DIM: int = 768  # src/retrievers.py:13
logger = logging.getLogger(__name__)

_GID_BITS = 47
_GID_MASK = (1 << _GID_BITS) - 1


# --------------------------------------------------------------------------------------------
# packed candidates, host side (numpy). Same bit layout as csrc/common.h pack_candidate().
# --------------------------------------------------------------------------------------------
def _order_key16(score_bits: np.ndarray) -> np.ndarray:
    h = score_bits.astype(np.int64) & 0xFFFF
    h = np.where((h & 0x7FFF) == 0, 0, h)
    return np.where(h & 0x8000, 0xFFFF - h, h | 0x8000)


def pack_candidates_host(scores_f16: np.ndarray, idx: np.ndarray, id_mul: int, id_add: int) -> np.ndarray:
    """(score fp16, local row) -> int64 packed candidate; rows with idx < 0 pack to 0."""
    key = _order_key16(scores_f16.view(np.uint16))
    gid = idx.astype(np.int64) * id_mul + id_add
    packed = (key << _GID_BITS) | (_GID_MASK - (gid & _GID_MASK))
    return np.where(idx < 0, 0, packed).astype(np.int64)


def unpack_candidates_host(packed: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """int64 packed -> (score fp16, global id); packed == 0 -> (-inf, -1)."""
    key = (packed >> _GID_BITS) & 0xFFFF
    bits = np.where(key & 0x8000, key & 0x7FFF, 0xFFFF - key).astype(np.uint16)
    gid = _GID_MASK - (packed & _GID_MASK)
    empty = packed == 0
    bits = np.where(empty, np.uint16(0xFC00), bits).astype(np.uint16)
    gid = np.where(empty, -1, gid).astype(np.int64)
    return bits.view(np.float16), gid


def merge_packed_host(gathered: np.ndarray, k: int) -> np.ndarray:
    """(W, B, k) packed -> (B, k): the k largest per query, descending (index.py:151 over W*k)."""
    W, B, kk = gathered.shape
    flat = np.ascontiguousarray(gathered.transpose(1, 0, 2)).reshape(B, W * kk)
    return -np.sort(-flat, axis=1, kind="stable")[:, :k]


class _DocMap(dict):
    """the doc_map this class builds (init_embeddings / load_index): a plain dict for every caller, plus a counter of in-place changes so that
    `index.doc_map[i] = passage` is seen by the id -> passage mirror of `_docs_of_rows` (the reference re-reads doc_map[x] on every search)"""
    __slots__ = ("edits",)

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.edits = 0

    def __setitem__(self, key, value):
        self.edits += 1
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self.edits += 1
        dict.__delitem__(self, key)

    def update(self, *a, **kw):
        self.edits += 1
        dict.update(self, *a, **kw)

    def pop(self, *a):
        self.edits += 1
        return dict.pop(self, *a)

    def popitem(self):
        self.edits += 1
        return dict.popitem(self)

    def clear(self):
        self.edits += 1
        dict.clear(self)

    def setdefault(self, *a):
        self.edits += 1
        return dict.setdefault(self, *a)

    def __ior__(self, other):                   # (`dm |= other` does not go through update())
        self.edits += 1
        return dict.__ior__(self, other)

    def __reduce__(self):                       # pickles (and deep-copies) as a plain dict's content
        return (_DocMap, (dict(self),))


class HipDistributedIndex(object):
    def __init__(self, certify_every: int = 64, exchange: str = "rccl"):
        """exchange: how the ranks' packed winners meet in a distributed search_knn: "rccl" = one all_gather_into_tensor + the W*k -> k
        merge (the default); "peer" = every rank writes them into its peers' hipIpc-mapped exchange buffers and the merge waits for the
        tags (dist_utils.PeerExchange; experimental: back on "rccl" for good if the buffers cannot be mapped; a late peer raises).
        certify_every: every that-many-th search runs the scan that measures every row's norm itself (the C-ABI's default mode,
        ~5 % slower) instead of trusting the bound taken when the slab last changed -- the net under writers torch's version counter
        cannot see (`.data`, a numpy / DLPack alias, a raw pointer). 1 = every search certifies, 0 = never (trust the counter alone)."""
        self.certify_every = int(certify_every)
        assert exchange in ("rccl", "peer"), exchange
        self.exchange = exchange
        self._peer_xchg = None
        self._since_certified = 0
        self.embeddings = None          # (d, N) fp16 view of the slab, like the reference
        self.doc_map = dict()
        self.is_in_gpu = True
        self._slab = None               # (N, d) fp16, row-major, contiguous
        self._pmax: Optional[float] = None     # certified upper bound on row norms (None = unknown)
        self._pmax_version = None              # torch version counter of the slab when _pmax was measured
        self._ws = None
        self._ws_bytes_cache = {}
        self._host_out = None                # pinned D2H buffer of the search results (reused)
        self._doc_arr, self._doc_arr_tag = None, None     # doc_map mirrored as a numpy object array (see _docs_of_rows)
        self._min_shard_rows = None          # the smallest shard of the job (collective, taken at the first search after the slab was (re)bound)
        self._ws_exact = None
        self._last_packed = None
        self._gid_mode = "round_robin"  # how local rows map to global passage ids
        self._gid_offset = 0
        self._gid_bounds = None         # contiguous mode: cumulative shard sizes of all ranks
        self._passage_store = None      # optional node-local PassageStore (attach_passage_store)
        self._warned_exact = set()
        self.last_search_stats = {}

    def attach_passage_store(self, store) -> None:
        """Resolve the winners' passages from a node-local `passage_store.PassageStore` (keyed by global passage id)
        instead of exchanging them between ranks: search_knn then has no text collective.
        The store must have been built in this index's global-id order: `PassageStore.iter_jsonl` for passages loaded
        round-robin by `index_io.load_passages`, `PassageStore.iter_saved_index` for an index loaded with `load_index`.
        Collective (every rank attaches): the store must hold exactly as many passages as all shards together."""
        both = dist_utils.all_gather_object((len(self.doc_map), -1 if self._slab is None else int(self._slab.shape[0])))
        total = sum(int(n) for n, _ in both)
        if all(int(r) >= 0 for _, r in both):          # (decided from what EVERY rank reported: all ranks take it, or none does)
            self._min_shard_rows = min(int(r) for _, r in both)      # the shard sizes of the job, for free: search_knn's collective range check
        if len(store) != total:
            raise ValueError(f"passage store holds {len(store)} passages, the index {total}: it was built from another corpus")
        self._passage_store = store
        self._store_edits = getattr(self.doc_map, "edits", 0)      # (see search_knn: in-place doc_map edits made after this point win for this rank's own rows)

    # ------------------------------------------------------------------ storage
    def _device(self):
        if self.is_in_gpu and torch.cuda.is_available():
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def _set_slab(self, slab: torch.Tensor):
        assert slab.dtype == torch.float16 and slab.dim() == 2 and slab.is_contiguous()
        self._slab = slab
        self.embeddings = slab.T
        self._pmax = None
        self._pmax_version = None
        self._min_shard_rows = None
        self._doc_arr, self._doc_arr_tag = None, None     # (a new slab comes with a new doc_map: never serve the old mirror -- ADVICE r04)

    def init_embeddings(self, passages, dim: Optional[int] = EMBEDDINGS_DIM):
        """index.py:48-53 — allocate a zeroed slab for `passages` and the local doc map."""
        self.doc_map = _DocMap(enumerate(passages))
        self._set_slab(torch.zeros((len(passages), dim), dtype=torch.float16, device=self._device()))
        self._gid_mode = "round_robin"   # src/index_io.py:41: global line c -> rank c % W, slot c // W

    def _check_slab(self):
        # callers may only write through `embeddings[...] = ...`; if somebody rebinds the
        # attribute (as the reference's own load_index does), adopt the new tensor
        e = self.embeddings
        assert e is not None
        if self._slab is None or e.data_ptr() != self._slab.data_ptr() or tuple(e.shape) != (self._slab.shape[1], self._slab.shape[0]):
            self._set_slab(e.T.contiguous() if not e.T.is_contiguous() else e.T)

    # ------------------------------------------------------------------ persistence
    # On-disk format of the reference (src/index.py:55-111, preprocessing/download_index.py:12-14), which prebuilt Atlas
    # indices use: `total_saved_shards` pairs of files, rank r owning the ids [r*S/W, (r+1)*S/W):
    #     embeddings.{id}.pt   torch.save of a contiguous (d, n_id) fp16 tensor
    #     passages.{id}.pt     pickle of the list of the n_id passage dicts
    # A rank's rows are cut into runs of ceil(n / (S/W)) rows. The slab is (N, d), so a shard is transposed on the device
    # on its way out / in; the file contents are the reference's.
    def _get_saved_embedding_path(self, save_dir: str, shard: int) -> str:
        return os.path.join(save_dir, "embeddings.%d.pt" % shard)

    def _get_saved_passages_path(self, save_dir: str, shard: int) -> str:
        return os.path.join(save_dir, "passages.%d.pt" % shard)

    @staticmethod
    def _owned_shard_ids(total_saved_shards: int) -> range:
        world, rank = dist_utils.get_world_size(), dist_utils.get_rank()
        assert total_saved_shards % world == 0, "N workers must be a multiple of shards to save"
        per_rank = total_saved_shards // world
        return range(rank * per_rank, (rank + 1) * per_rank)

    @staticmethod
    def _row_runs(n_rows: int, n_files: int) -> List[Tuple[int, int]]:
        """[start, end) of the rows each of this rank's files holds; files past the last row hold nothing."""
        run = -(-n_rows // n_files) if n_rows else 0
        return [(min(j * run, n_rows), min((j + 1) * run, n_rows)) for j in range(n_files)]

    def save_index(self, path: str, total_saved_shards: int, overwrite_saved_passages: bool = False) -> None:
        """Embeddings are always rewritten; a passages file is kept if it already exists unless `overwrite_saved_passages`
        (index.py:80-83). Unlike the reference, a rank with fewer rows than files (or none: the reference divides by a zero
        step there) still writes all of its files, the surplus ones empty, so that `load_index` finds every id."""
        assert self.embeddings is not None
        self._check_slab()
        ids = self._owned_shard_ids(total_saved_shards)
        n_rows = int(self._slab.shape[0])
        assert n_rows == len(self.doc_map), len(self.doc_map)
        for shard_id, (lo, hi) in zip(ids, self._row_runs(n_rows, len(ids))):
            p_file = self._get_saved_passages_path(path, shard_id)
            if overwrite_saved_passages or not os.path.exists(p_file):
                with open(p_file, "wb") as fobj:
                    pickle.dump([self.doc_map[r] for r in range(lo, hi)], fobj, protocol=pickle.HIGHEST_PROTOCOL)
            block = self._slab[lo:hi].T.contiguous().cpu()         # (d, n): transposed where the slab lives, then one D2H copy
            torch.save(block, self._get_saved_embedding_path(path, shard_id))

    def load_index(self, path: str, total_saved_shards: int):
        """Reads this rank's files (written by this class or by the reference) into one pre-allocated slab: every (d, n) block is
        moved to the device as it is and transposed THERE into its rows (peak device memory = slab + one block; nothing is
        transposed or concatenated on the host)."""
        ids = self._owned_shard_ids(total_saved_shards)
        chunks = []
        for shard_id in ids:
            with open(self._get_saved_passages_path(path, shard_id), "rb") as fobj:
                chunks.append(pickle.load(fobj))
        n_rows = sum(len(c) for c in chunks)
        dev = self._device()
        slab, dim, row = None, EMBEDDINGS_DIM, 0
        for shard_id, chunk in zip(ids, chunks):
            block = torch.load(self._get_saved_embedding_path(path, shard_id), map_location="cpu")
            assert block.dim() == 2 and block.shape[1] == len(chunk), (tuple(block.shape), len(chunk))
            if slab is None:
                dim = int(block.shape[0])
                slab = torch.empty((n_rows, dim), dtype=torch.float16, device=dev)
            assert block.shape[0] == dim
            if len(chunk):
                slab[row: row + len(chunk)].copy_(block.to(device=dev, dtype=torch.float16).T)
            row += len(chunk)
        if slab is None:
            slab = torch.empty((0, dim), dtype=torch.float16, device=dev)
        self.doc_map = _DocMap(enumerate(p for chunk in chunks for p in chunk))
        self._set_slab(slab)
        # saved shards are contiguous runs of passages: global id = offset of this rank + row
        self._gid_mode = "contiguous"
        sizes = dist_utils.all_gather_object(n_rows)
        self._min_shard_rows = min(int(s) for s in sizes)          # (search_knn's collective range check, for free)
        self._gid_bounds = np.cumsum([0] + [int(s) for s in sizes])
        self._gid_offset = int(self._gid_bounds[dist_utils.get_rank()])

    # ------------------------------------------------------------------ global ids
    def _gid_params(self) -> Tuple[int, int]:
        if self._gid_mode == "round_robin":
            return dist_utils.get_world_size(), dist_utils.get_rank()
        return 1, self._gid_offset

    def _gid_owner(self, gid: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """global id -> (owner rank, local row)."""
        if self._gid_mode == "round_robin":
            W = dist_utils.get_world_size()
            return gid % W, gid // W
        owner = np.searchsorted(self._gid_bounds, gid, side="right") - 1
        return owner, gid - self._gid_bounds[owner]

    # ------------------------------------------------------------------ local search (HIP)
    def _workspace(self, nbytes: int, exact: bool = False) -> torch.Tensor:
        attr = "_ws_exact" if exact else "_ws"
        ws = getattr(self, attr)
        if ws is None or ws.numel() < nbytes or ws.device != self._slab.device:
            # a scan workspace is zero-filled once: its head holds state that lives across calls (include/atlas_hip.h); the exact
            # path's holds nothing between calls (a multi-GiB memset on the first tie-heavy query would be pure cost)
            alloc = torch.empty if exact else torch.zeros
            ws = alloc(int(nbytes), dtype=torch.uint8, device=self._slab.device)
            setattr(self, attr, ws)
        return ws

    def _require_gpu(self):
        if self._slab is None:
            raise _lib.AtlasHipError("index has no embeddings")
        if not self._slab.is_cuda:
            raise _lib.AtlasHipError(
                "HipDistributedIndex computes on an MI355X only: the slab is on the CPU and there is no "
                "CPU fallback (is_in_gpu=False or no GPU visible)."
            )
        return _lib.lib()

    def _slab_version(self) -> Optional[int]:
        """torch's in-place version counter of the slab: every write through torch (`index.embeddings[:, a:b] = X.T` of atlas.py:79
        included -- views share the counter) bumps it; the HIP encoder's direct row writes bump it too (Contriever.embed_into).
        None for a tensor without a counter (created under torch.inference_mode()): such a slab is re-certified by every scan."""
        try:
            return int(self._slab._version)
        except RuntimeError:
            return None

    def invalidate_pmax(self) -> None:
        """for writers that reach the slab's memory without torch (a raw pointer from another library): the next search measures
        the row norms again"""
        self._pmax = None
        self._pmax_version = None

    def slab_pmax(self) -> float:
        """Max L2 row norm of the slab, rounded up (one streaming pass): a certified bound for ATLAS_SCAN_TRUST_PMAX."""
        L = self._require_gpu()
        N, d = self._slab.shape
        out = torch.zeros(1, dtype=torch.float32, device=self._slab.device)
        stream = torch.cuda.current_stream(self._slab.device).cuda_stream
        _lib.check(L.atlas_slab_pmax(self._slab.data_ptr(), N, d, out.data_ptr(), stream), "atlas_slab_pmax")
        return float(out.item())

    def _exact_topk(self, q: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """MFMA-free exact path (any d, k <= 2048): same canonical result as the scan."""
        L = self._require_gpu()
        N, d = self._slab.shape
        B = q.shape[0]
        code = _lib.torch_dtype_code(q.dtype)
        if code is None:
            q, code = q.float(), _lib.DT_F32
        q = q.contiguous()
        ws = self._workspace(L.atlas_exact_topk_workspace_bytes(N, B, d, k), exact=True)
        out_s = torch.empty((B, k), dtype=torch.float16, device=q.device)
        out_i = torch.empty((B, k), dtype=torch.int64, device=q.device)
        stream = torch.cuda.current_stream(q.device).cuda_stream
        _lib.check(L.atlas_exact_topk(q.data_ptr(), code, self._slab.data_ptr(), N, B, d, k, out_s.data_ptr(),
                                      out_i.data_ptr(), ws.data_ptr(), ws.numel(), stream), "atlas_exact_topk")
        return out_s, out_i

    def _local_topk(self, q: torch.Tensor, k: int, pack=None):
        """Fused scan + top-k over this shard. Returns device (scores fp16 [B,k], rows int64 [B,k])
        and their host copies (numpy), after the status word has been checked. The host copies are VIEWS OF A REUSED PINNED BUFFER,
        valid until the next `_local_topk` of this index (every caller in this file converts them to lists / packs them at once);
        a caller that keeps them across searches copies them (tests/test_index_host.py::test_host_results_do_not_alias_across_searches
        pins what search_knn hands out).
        pack = (id_mul, id_add): the merge kernel also emits the winners as cross-shard packed candidates (one launch less in front of
        the all-gather); they are left in `self._last_packed` ((B, k) int64, device), or None when a query took another path.

        Protocol (include/atlas_hip.h): the scan certifies its pruning margin with an upper bound
        on the row norms; if a larger row is met the call is repeated once with the measured
        bound; queries whose candidate band overflowed (mass ties) are redone on the exact path.
        """
        L = self._require_gpu()
        N, d = self._slab.shape
        B = q.shape[0]
        if q.device != self._slab.device:
            q = q.to(self._slab.device)
        self._last_packed = None
        if B == 0:      # an empty batch (atlas.py:106 builds one; ranks with no queries still take part in the collectives)
            s = torch.empty((0, k), dtype=torch.float16, device=q.device)
            i = torch.empty((0, k), dtype=torch.int64, device=q.device)
            self.last_search_stats = {"path": "empty"}
            return s, i, s.cpu().numpy(), i.cpu().numpy()
        if d != _lib.D_FAST or k > _lib.K_FAST_MAX:
            if k > _lib.K_EXACT_MAX:
                raise _lib.AtlasHipError(f"topk={k} exceeds the supported maximum {_lib.K_EXACT_MAX}")
            if (d, k) not in self._warned_exact:       # same canonical result, but one fp64 slab pass per 8 queries instead of the MFMA scan
                self._warned_exact.add((d, k))
                logger.warning("topk=%d / d=%d is outside the fused scan (d == %d, k <= %d): whole batches take the exact path "
                               "(~20 ms per 8 queries at 32M rows)", k, d, _lib.D_FAST, _lib.K_FAST_MAX)
            s, i = self._exact_topk(q, k)
            self.last_search_stats = {"path": "exact"}
            return s, i, s.cpu().numpy(), i.cpu().numpy()
        code = _lib.torch_dtype_code(q.dtype)
        if code is None:
            q, code = q.float(), _lib.DT_F32
        q = q.contiguous()
        # The certified error margin needs an upper bound of the row norms. It is measured once per STATE of the slab (one streaming
        # pass, atlas_slab_pmax) and trusted for as long as torch's version counter says nothing wrote to the slab since; the scan then
        # skips its own per-row measurement (4.6 % of its time). Without a counter every scan certifies the bound itself.
        version = self._slab_version()
        if version is None:
            call_flags = 0
            if self._pmax is None:
                self._pmax = self.slab_pmax()
        else:
            call_flags = _lib.SCAN_TRUST_PMAX
            if self._pmax is None or self._pmax_version != version:
                self._pmax = self.slab_pmax()
                self._pmax_version = version
                self._since_certified = 0
            # ... and every certify_every-th search certifies anyway: a write that went around the counter is found by the scan itself
            self._since_certified += 1
            if self.certify_every > 0 and self._since_certified >= self.certify_every:
                call_flags = 0
        key = (N, B, d, k)
        ws_bytes = self._ws_bytes_cache.get(key)
        if ws_bytes is None:                               # (the plan arithmetic behind it is ~20 us of host time per call)
            ws_bytes = self._ws_bytes_cache[key] = int(L.atlas_scan_topk_workspace_bytes(N, B, d, k))
        ws = self._workspace(ws_bytes)
        # one output buffer -> one D2H copy: [status int32 | scores fp16 | rows int64]
        n_st = _lib.STATUS_HEADER + B
        off_s = (n_st * 4 + 15) // 16 * 16
        off_i = (off_s + B * k * 2 + 15) // 16 * 16
        total = off_i + B * k * 8
        out = torch.empty(total, dtype=torch.uint8, device=q.device)
        stream = torch.cuda.current_stream(q.device).cuda_stream
        base = out.data_ptr()
        packed = torch.empty((B, k), dtype=torch.int64, device=q.device) if pack is not None else None
        id_mul, id_add = pack if pack is not None else (1, 0)
        reruns = 0
        while True:
            rc = L.atlas_scan_topk_pack(q.data_ptr(), code, self._slab.data_ptr(), N, B, d, k, float(self._pmax),
                                        base + off_s, base + off_i, base, ws.data_ptr(), ws.numel(), stream, None, None, call_flags,
                                        int(id_mul), int(id_add), packed.data_ptr() if packed is not None else None)
            if rc != 0:
                # a launch that failed half-way may have left the workspace's per-call state (tile-pool ticket, flags) behind: the
                # next search starts from a fresh zero-filled one
                self._ws = None
            _lib.check(rc, "atlas_scan_topk")
            # ONE pinned D2H of the whole result (a pageable `.cpu()` stages through a bounce buffer and allocates per call); the pinned
            # buffer is reused: what is handed out below are copies or lists made from it before the next search
            if self._host_out is None or self._host_out.numel() < total:
                self._host_out = torch.empty(max(total, 1 << 16), dtype=torch.uint8, pin_memory=True)
            hbuf = self._host_out[:total]
            hbuf.copy_(out, non_blocking=True)
            torch.cuda.current_stream(q.device).synchronize()
            host = hbuf.numpy()
            st = host[: n_st * 4].view(np.int32)
            flags = int(st[_lib.ST_FLAGS])
            pmax_seen = float(st[_lib.ST_PMAX_BITS : _lib.ST_PMAX_BITS + 1].view(np.float32)[0])
            if flags & _lib.F_PMAX_VIOLATION and reruns < 2:
                if call_flags & _lib.SCAN_TRUST_PMAX:
                    # the merge met a rescored row longer than the trusted bound: something wrote to the slab behind torch's back.
                    # Its norm is only a lower bound of the maximum -- the repeat certifies for itself, from a fresh measurement
                    logger.warning("a slab row is longer than the certified bound (%.4g > %.4g): the slab was written without a version "
                                   "bump; re-certifying (see invalidate_pmax)", pmax_seen, self._pmax)
                    self._pmax = self.slab_pmax()
                    call_flags = 0
                else:
                    self._pmax = pmax_seen    # the scan measured the true maximum: certified on the re-run
                reruns += 1
                continue
            break
        if not (call_flags & _lib.SCAN_TRUST_PMAX):
            self._since_certified = 0
        if flags & (_lib.F_PMAX_VIOLATION | _lib.F_EPS_VIOLATION):
            raise _lib.AtlasHipError(f"scan could not certify its result (flags={flags}); this is a bug")
        self._pmax = pmax_seen if pmax_seen > 0 else self._pmax
        scores = out[off_s : off_s + B * k * 2].view(torch.float16).view(B, k)
        rows = out[off_i : off_i + B * k * 8].view(torch.int64).view(B, k)
        h_scores = host[off_s : off_s + B * k * 2].view(np.float16).reshape(B, k)      # (views of the pinned buffer: valid until the next search)
        h_rows = host[off_i : off_i + B * k * 8].view(np.int64).reshape(B, k)
        n_fb = 0
        if flags & _lib.F_FALLBACK:
            sel = np.nonzero(st[_lib.STATUS_HEADER :] != 0)[0]
            n_fb = len(sel)
            sel_t = torch.as_tensor(sel, device=q.device, dtype=torch.int64)
            es, ei = self._exact_topk(q.index_select(0, sel_t), k)
            scores.index_copy_(0, sel_t, es)
            rows.index_copy_(0, sel_t, ei)
            h_scores[sel] = es.cpu().numpy()
            h_rows[sel] = ei.cpu().numpy()
        self._last_packed = packed if n_fb == 0 else None        # (exact-path rows are packed by the caller)
        self.last_search_stats = {
            "path": "scan", "reruns": reruns, "fallback_queries": n_fb, "pmax": self._pmax, "pmax_trusted": bool(call_flags & _lib.SCAN_TRUST_PMAX),
            "candidates": int(st[_lib.ST_N_CANDIDATES]), "rescored": int(st[_lib.ST_N_RESCORED]),
            "max_err_over_eps": float(st[_lib.ST_MAXERR_BITS : _lib.ST_MAXERR_BITS + 1].view(np.float32)[0]),
            "plan": _lib.decode_plan(st[_lib.ST_PLAN]),
        }
        return scores, rows, h_scores, h_rows

    def _compute_scores_and_indices(self, allqueries: torch.Tensor, topk: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """index.py:113-120 — (B, d) queries -> (scores fp16 (B, topk) desc, shard-local indices int64 (B, topk))."""
        self._check_slab()
        if topk > self._slab.shape[0]:
            # the reference's torch.topk raises here; keep that contract
            raise RuntimeError(f"selected index k out of range (topk={topk} > {self._slab.shape[0]} passages in shard)")
        scores, rows, _, _ = self._local_topk(allqueries, topk)
        return scores, rows

    # ------------------------------------------------------------------ search (index.py:122-157)
    @torch.no_grad()
    def search_knn(self, queries, topk):
        """
        Conducts exhaustive search of the k-nearest neighbours using the inner product metric.
        Collective: every rank calls it the same number of times with the same topk.
        """
        self._check_slab()
        # `topk > rows of a shard` is the reference's torch.topk error (index.py:118) -- raised there, and here until round 4, by whichever
        # ranks own a short shard AFTER the query collective, while the others went on into the next collective and hung (shards that differ
        # by one row are the normal case: N % W != 0). The smallest shard of the job is taken once per slab (one all_gather_object at the
        # first search after init_embeddings / load_index, both of which re-bind the slab) and every rank raises, or none, BEFORE any
        # collective of the search.
        if self._min_shard_rows is None:
            self._min_shard_rows = min(int(n) for n in dist_utils.all_gather_object(int(self._slab.shape[0])))
        if topk > self._min_shard_rows:
            raise RuntimeError(f"selected index k out of range (topk={topk} > {self._min_shard_rows} passages in the smallest shard; "
                               f"this rank's shard holds {self._slab.shape[0]})")
        allqueries, allsizes = dist_utils.all_gather_queries(queries)
        bounds = np.cumsum([0] + list(allsizes))
        distributed = dist_utils.is_initialized()
        scores_d, rows_d, scores, rows = self._local_topk(allqueries, topk, pack=self._gid_params() if distributed else None)
        if not distributed:
            return self._docs_of_rows(rows), scores.astype(np.float64).tolist()

        rank = dist_utils.get_rank()
        id_mul, id_add = self._gid_params()
        packed = self._last_packed                                                       # straight from the merge kernel, or ...
        if packed is None:
            packed = self._pack(scores_d, rows_d, scores, rows, id_mul, id_add)         # (B, k) int64, device
        merged = self._peer_merge(packed, topk) if self.exchange == "peer" and packed.is_cuda else None
        if merged is None:
            gathered = dist_utils.all_gather_packed(packed)                              # (W, B, k): ONE collective
            merged = self._merge(gathered, topk)                                         # (B, k) numpy, W*k -> k per query
        m_scores, m_gid = unpack_candidates_host(merged)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        if self._passage_store is not None:
            # node-local passage store: ids resolve locally, no text collective at all
            mine_g, mine_s = m_gid[lo:hi], m_scores[lo:hi]
            if bool((mine_g >= 0).all()):                  # (the common case: every query has k real winners -- lists built in C)
                docs = [self._passage_store.get_many(row) for row in mine_g]
                out_scores = mine_s.astype(np.float64).tolist()
            else:
                docs = [self._passage_store.get_many(row[row >= 0]) for row in mine_g]
                out_scores = [[float(s) for s, g in zip(srow, grow) if g >= 0] for srow, grow in zip(mine_s, mine_g)]
            if getattr(self.doc_map, "edits", 0) != getattr(self, "_store_edits", 0):
                # the store is a snapshot of the corpus taken when it was built; `doc_map` entries edited in place since then (the counter
                # of _DocMap) win for the winners that live in THIS rank's shard -- no collective involved, so a rank that edited and one that
                # did not stay in step. Edits made on OTHER ranks are not visible through a store: re-attach one built from the new text.
                for drow, grow in zip(docs, mine_g):
                    g = grow[grow >= 0]
                    owner, local = self._gid_owner(g)
                    for j in np.nonzero(owner == rank)[0].tolist():
                        drow[j] = self.doc_map[int(local[j])]
            return docs, out_scores
        owner, local = self._gid_owner(np.maximum(m_gid, 0))
        # passage text: for every rank, the winners of ITS queries that live in this shard (k per query and destination, not
        # W*k, and nothing a rank did not ask for), in one personalised exchange
        W = dist_utils.get_world_size()
        mine = (owner == rank) & (m_gid >= 0)
        outbox = []
        for dst in range(W):
            rows = slice(int(bounds[dst]), int(bounds[dst + 1]))
            sel = mine[rows]
            outbox.append({int(g): self.doc_map[int(l)] for g, l in zip(m_gid[rows][sel], local[rows][sel])})
        table = {}
        for part in dist_utils.exchange_objects(outbox):
            table.update(part)
        docs = [[table[int(g)] for g in m_gid[b] if g >= 0] for b in range(lo, hi)]
        out_scores = [[float(s) for s, g in zip(m_scores[b], m_gid[b]) if g >= 0] for b in range(lo, hi)]
        return docs, out_scores

    def _docs_of_rows(self, rows: np.ndarray):
        """[b][k] shard-local rows -> the passages (the SAME dict objects `doc_map[row]` returns, src/index.py:131). A dict doc_map with
        the dense keys 0..n-1 the reference builds (index.py:47, :106) is mirrored ONCE per (dict identity, length) into a numpy object
        array: 2 560 lookups become one fancy-indexing take + tolist() in C (~25 us instead of ~250 us per batch of 64 x 40). In-place
        edits of a doc_map this class built (`index.doc_map[i] = p`) are seen through `_DocMap.edits`; a caller that assigned its OWN plain
        dict and swaps entries of it in place (same length) calls `index.invalidate_doc_cache()`; anything that is not a dict is looked up
        entry by entry."""
        dm = self.doc_map
        if isinstance(dm, dict) and len(dm) > 0:
            # the tag holds the mirrored dict ITSELF (compared with `is`: a freed dict's id() may be reused), its length and -- for the
            # dicts this class builds -- its edit counter
            tag = self._doc_arr_tag
            edits = getattr(dm, "edits", 0)
            if tag is None or tag[0] is not dm or tag[1] != len(dm) or tag[2] != edits:
                arr = None
                n = len(dm)
                if 0 in dm and (n - 1) in dm:
                    try:
                        arr = np.empty(n, dtype=object)
                        arr[:] = [dm[i] for i in range(n)]          # KeyError: the keys are not 0..n-1
                    except KeyError:
                        arr = None
                self._doc_arr, self._doc_arr_tag = arr, (dm, n, edits)
            if self._doc_arr is not None:
                return self._doc_arr[rows].tolist()
        return [[dm[x] for x in sample] for sample in rows.tolist()]     # (tolist() converts in C: half the host time of per-element int())

    def invalidate_doc_cache(self) -> None:
        """after replacing entries of `doc_map` IN PLACE (same dict, same length): the id -> passage mirror is rebuilt at the next search"""
        self._doc_arr, self._doc_arr_tag = None, None

    def _peer_merge(self, packed: torch.Tensor, k: int) -> Optional[np.ndarray]:
        """the "peer" exchange of one search; None = use the collective (the set-up failed -- on every rank alike -- and the index is
        back on "rccl" for good). A peer that is LATE is an error, not a fallback: agreeing on a fallback would cost the very collective
        the exchange is there to avoid."""
        B = int(packed.shape[0])
        if B == 0:
            return None
        if self._peer_xchg is None or self._peer_xchg.slot_entries < B * k:
            if self._peer_xchg is not None:
                self._peer_xchg.close()
                self._peer_xchg = None
            try:
                self._peer_xchg = dist_utils.PeerExchange(slot_entries=max(B * k, 64 * 256))
            except _lib.AtlasHipError as e:
                warnings.warn(f"{e}; using the collective")
                self.exchange = "rccl"
                return None
        out = self._peer_xchg.exchange(packed, k)
        if out is None:
            # (deliberately not a per-rank fallback: the peers that were served would go on to the NEXT collective while this rank repeated
            #  this one -- the job would hang instead of failing; wait_ms is seconds, a peer that late is gone)
            raise _lib.AtlasHipError(f"peer exchange: a rank did not deliver its candidates within {self._peer_xchg.wait_ms} ms")
        return out.cpu().numpy()

    def _pack(self, scores_d, rows_d, scores_h, rows_h, id_mul, id_add) -> torch.Tensor:
        if scores_d.numel() == 0:                       # every rank's batch is empty: nothing to launch
            return torch.empty(scores_d.shape, dtype=torch.int64, device=scores_d.device)
        if scores_d.is_cuda:
            L = _lib.lib()
            out = torch.empty(scores_d.shape, dtype=torch.int64, device=scores_d.device)
            stream = torch.cuda.current_stream(scores_d.device).cuda_stream
            _lib.check(L.atlas_pack_candidates(scores_d.data_ptr(), rows_d.data_ptr(), scores_d.numel(), id_mul, id_add,
                                               out.data_ptr(), stream), "atlas_pack_candidates")
            return out
        # process groups on CPU tensors (gloo): same bit layout, host arithmetic
        return torch.from_numpy(pack_candidates_host(scores_h, rows_h, id_mul, id_add))

    def _merge(self, gathered: torch.Tensor, k: int) -> np.ndarray:
        """W*k -> k per query under the canonical order (replaces index.py:151)."""
        W, B, kk = gathered.shape
        if B == 0:
            return np.empty((0, k), dtype=np.int64)
        if gathered.is_cuda and W * kk <= 8192:
            L = _lib.lib()
            out = torch.empty((B, k), dtype=torch.int64, device=gathered.device)
            stream = torch.cuda.current_stream(gathered.device).cuda_stream
            _lib.check(L.atlas_merge_packed(gathered.data_ptr(), W, B, k, out.data_ptr(), stream), "atlas_merge_packed")
            return out.cpu().numpy()
        return merge_packed_host(gathered.cpu().numpy(), k)

    def is_index_trained(self) -> bool:
        return True

    def train_index(self):  # never called for a flat index (atlas.py:86-88)
        pass
