"""Rank helpers + the two collectives the search path needs.

Replaces the hot-path use of src/dist_utils.py (varsize_all_gather :46-69, varsize_gather :72-99,
get_varsize :102-113). The reference issues 3 + 4*W collectives per search_knn call (SURVEY §2.3
C1-C5); here a search is: ONE fixed-size all_gather of [batch size | padded fp16 queries] (C1+C2 in one collective and one host
sync, C3 is redundant and dropped), one all_gather of packed (score,id) candidates (replaces C4+C5), and a
personalised exchange of the winning passages (every rank receives the k winners of ITS OWN queries
only; no text collective at all with a node-local passage store attached).

`backend="nccl"` on PyTorch-ROCm is RCCL; tensors handed to a collective live on the device the
process group's backend expects (cuda for nccl, cpu for gloo) — callers pass device tensors.
"""
from typing import List, Tuple

import torch
from typing import Optional
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_rank() -> int:
    return dist.get_rank() if is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if is_initialized() else 1


def barrier() -> None:
    if is_initialized():
        dist.barrier()


# rows per rank of the fixed-size query collective, PER PROCESS GROUP: {group key: [cap, consecutive calls that would have fitted 64 rows]}.
# It grows (on every rank alike: same gathered headers) when a batch exceeds it and falls back to 64 after _CAP_DECAY_CALLS small calls in a row,
# so that one large evaluation batch does not make every later 1-query search gather W x (cap + 1) x 768 halfs for good.
_QUERY_CAP_MIN = 64
_CAP_DECAY_CALLS = 16
_query_caps = {}


@torch.no_grad()
def all_gather_queries(queries: torch.Tensor) -> Tuple[torch.Tensor, List[int]]:
    """(b_r, d) queries of any float dtype -> ((B, d) fp16 of all ranks in rank order, [b_0..b_{W-1}]).

    ONE collective and ONE host sync per call (the reference: a size all-gather with W `.item()` syncs, then the padded fp32 queries,
    src/dist_utils.py:46-69): every rank sends a fixed-size block [header row | cap query rows]; the header carries its batch size as
    an int32. `cap` is process state that only grows: a batch beyond it is noticed by every rank in the same gathered headers, and all
    of them repeat the call once with the larger block. fp16 on the wire: the scan casts with `.half()` anyway (src/index.py:117), so
    gathering the fp32 originals moves twice the bytes for the same result.
    """
    q16 = queries.to(torch.float16)
    if not is_initialized():
        return q16, [q16.shape[0]]
    W = dist.get_world_size()
    b, d = q16.shape
    state = _query_caps.setdefault((id(dist.group.WORLD), W), [_QUERY_CAP_MIN, 0])
    while True:
        cap = state[0]
        block = torch.zeros((cap + 1, d), dtype=torch.float16, device=q16.device)
        block[0].view(torch.int32)[0] = b
        block[1 : 1 + min(b, cap)] = q16[:cap]
        out = torch.empty((W * (cap + 1), d), dtype=torch.float16, device=q16.device)
        dist.all_gather_into_tensor(out, block)
        out = out.view(W, cap + 1, d)
        sizes = out[:, 0].contiguous().view(torch.int32)[:, 0].tolist()          # the one host sync
        if max(sizes) <= cap:
            break
        state[0], state[1] = (max(sizes) + 63) // 64 * 64, 0                         # same decision on every rank: same headers
    # decay (every rank sees the same sizes, so every rank shrinks at the same call)
    if cap > _QUERY_CAP_MIN:
        state[1] = state[1] + 1 if max(sizes) <= _QUERY_CAP_MIN else 0
        if state[1] >= _CAP_DECAY_CALLS:
            state[0], state[1] = _QUERY_CAP_MIN, 0
    allq = torch.cat([out[r, 1 : 1 + n] for r, n in enumerate(sizes)], dim=0)
    return allq, [int(n) for n in sizes]


@torch.no_grad()
def all_gather_packed(packed: torch.Tensor) -> torch.Tensor:
    """(B, k) int64 packed candidates -> (W, B, k): ONE fixed-size collective, 8*B*k bytes per rank."""
    if not is_initialized():
        return packed.unsqueeze(0)
    W = dist.get_world_size()
    B, k = packed.shape
    out = torch.empty((W * B, k), dtype=packed.dtype, device=packed.device)   # concatenation along dim 0
    dist.all_gather_into_tensor(out, packed.contiguous())
    return out.view(W, B, k)


class PeerExchange:
    """The one-hop alternative to `all_gather_packed` + the W*k -> k merge (C-ABI atlas_xchg_*, include/atlas_hip.h): every rank writes its
    packed winners straight into a slot of every peer's exchange buffer (mapped through hipIpc handles exchanged once, here) and the merge
    kernel waits for the W tags. EXPERIMENTAL and off by default (`HipDistributedIndex(exchange="peer")`): it has never run across two
    devices. Setting it up is collective and fails on every rank or on none: a rank whose own buffer could not be created still takes part in
    both set-up collectives (with an empty handle) and the verdict is formed from what all ranks report. `exchange()` returns None when a peer
    was late -- `wait_ms` is therefore generous (a rank that re-ran its scan or took the exact path is seconds late, not dead): the caller
    treats a late peer as an error of the job, there is no per-rank fallback that would not desynchronise the collectives."""

    def __init__(self, slot_entries: int, wait_ms: int = 10000):
        import ctypes

        from . import _lib

        self.L, self.W, self.rank = _lib.lib(), dist.get_world_size(), dist.get_rank()
        self.slot_entries, self.wait_ms, self.tag = int(slot_entries), int(wait_ms), 0
        self.dev = torch.device("cuda", torch.cuda.current_device())
        own, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
        self.own, self._opened, failed = None, [], None
        rc = self.L.atlas_xchg_create(self.W, self.slot_entries, ctypes.byref(own), handle)
        if rc != 0:                                                     # (no raise here: the other ranks are about to enter the collectives below)
            failed = f"atlas_xchg_create on rank {self.rank}: {'ATLAS_E_' + str(rc) if rc < 0 else 'hipError_t ' + str(rc)}"
        else:
            self.own = own.value
        handles = [None] * self.W
        dist.all_gather_object(handles, bytes(handle.raw) if failed is None else b"")      # once per index: 64 bytes per rank
        self.peers = (ctypes.c_void_p * self.W)()
        if failed is None and any(len(h) != 64 for h in handles):
            failed = f"rank {self.rank}: a peer has no exchange buffer"
        for r, h in enumerate(handles):
            if failed is not None:
                break
            if r == self.rank:
                self.peers[r] = self.own
                continue
            p = ctypes.c_void_p()
            rc = self.L.atlas_xchg_open(h, ctypes.byref(p))
            if rc != 0:
                failed = f"atlas_xchg_open of rank {r}'s buffer on rank {self.rank}: hipError_t {rc}"
                break
            self.peers[r] = p.value
            self._opened.append(p.value)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # every buffer is mapped everywhere before the first push -- or nobody uses the exchange (the verdict is the same on every rank)
        verdicts = [None] * self.W
        dist.all_gather_object(verdicts, failed)
        if any(v is not None for v in verdicts):
            self.close()
            raise _lib.AtlasHipError("peer exchange unavailable: " + "; ".join(v for v in verdicts if v is not None))

    def exchange(self, packed: torch.Tensor, k: int) -> Optional[torch.Tensor]:
        """(B, k) int64 packed winners of this rank (B the same on all ranks) -> (B, k) merged, or None if a peer was late"""
        from . import _lib

        B = int(packed.shape[0])
        assert packed.is_cuda and packed.dtype == torch.int64 and B * k <= self.slot_entries
        self.tag += 1
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        packed = packed.contiguous()
        out = torch.empty((B, k), dtype=torch.int64, device=self.dev)
        self.status.zero_()
        _lib.check(self.L.atlas_xchg_push(packed.data_ptr(), B * k, self.peers, self.W, self.rank, self.slot_entries, self.tag, stream), "atlas_xchg_push")
        _lib.check(self.L.atlas_xchg_merge(self.own, self.W, B, k, self.slot_entries, self.tag, self.wait_ms, out.data_ptr(), self.status.data_ptr(), stream),
                   "atlas_xchg_merge")
        return out if int(self.status.item()) == 0 else None

    def close(self) -> None:
        for p in self._opened:
            self.L.atlas_xchg_close(p)
        self._opened = []
        if self.own:
            self.L.atlas_xchg_destroy(self.own)
            self.own = None


def _collective_device() -> torch.device:
    """where tensors handed to a collective must live: the GPU for nccl (= RCCL), the host for gloo"""
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def exchange_objects(per_dst: list) -> list:
    """Personalised exchange of python objects: rank r receives [per_dst[r] of rank 0, ..., of rank W-1].

    Default: ONE `all_gather_object` of the whole outbox, of which every rank keeps what was addressed to it -- the plainest object
    collective there is (W x the bytes of the personalised form: ~W x k short passages per query, host-pickled). It is the default
    because the personalised form below -- two `all_to_all_single` (sizes, then the pickled bytes with UNEVEN and possibly EMPTY splits,
    on device tensors under RCCL) -- has never run on RCCL with more than one rank (VERDICT r05 next #1c): the first multi-GPU job must
    not be the one to find out. `ATLAS_EXCHANGE=alltoall` (the same on every rank) selects it; tests/test_dist_gloo.py covers both forms
    over gloo. One-host jobs take neither: the node-local passage store resolves ids without a text collective (index_io)."""
    if not is_initialized():
        return [per_dst[0]]
    import os
    import pickle

    if os.environ.get("ATLAS_EXCHANGE", "allgather") != "alltoall":
        rank = dist.get_rank()
        return [outbox[rank] for outbox in all_gather_object(per_dst)]

    W, dev = dist.get_world_size(), _collective_device()
    assert len(per_dst) == W
    blobs = [pickle.dumps(o, protocol=pickle.HIGHEST_PROTOCOL) for o in per_dst]
    send_sizes = torch.tensor([len(b) for b in blobs], dtype=torch.int64, device=dev)
    recv_sizes = torch.empty(W, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_sizes, send_sizes)
    recv_sizes = recv_sizes.tolist()
    payload = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev) if sum(len(b) for b in blobs) else torch.empty(0, dtype=torch.uint8, device=dev)
    inbox = torch.empty(int(sum(recv_sizes)), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(inbox, payload, output_split_sizes=recv_sizes, input_split_sizes=[len(b) for b in blobs])
    raw = inbox.cpu().numpy().tobytes()
    out, at = [], 0
    for n in recv_sizes:
        out.append(pickle.loads(raw[at: at + n]))
        at += n
    return out


def all_gather_object(obj) -> list:
    if not is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
