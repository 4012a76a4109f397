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


_QUERY_CAP = 64          # rows per rank of the fixed-size query collective; grows (on every rank alike) when a batch exceeds it


@torch.no_grad()
def all_gather_queries(queries: torch.Tensor) -> Tuple[torch.Tensor, List[int]]:
    """(b_r, d) queries of any float dtype -> ((B, d) fp16 of all ranks in rank order, [b_0..b_{W-1}]).

    ONE collective and ONE host sync per call (the reference: a size all-gather with W `.item()` syncs, then the padded fp32 queries,
    src/dist_utils.py:46-69): every rank sends a fixed-size block [header row | cap query rows]; the header carries its batch size as
    an int32. `cap` is process state that only grows: a batch beyond it is noticed by every rank in the same gathered headers, and all
    of them repeat the call once with the larger block. fp16 on the wire: the scan casts with `.half()` anyway (src/index.py:117), so
    gathering the fp32 originals moves twice the bytes for the same result.
    """
    global _QUERY_CAP
    q16 = queries.to(torch.float16)
    if not is_initialized():
        return q16, [q16.shape[0]]
    W = dist.get_world_size()
    b, d = q16.shape
    while True:
        cap = _QUERY_CAP
        block = torch.zeros((cap + 1, d), dtype=torch.float16, device=q16.device)
        block[0].view(torch.int32)[0] = b
        block[1 : 1 + min(b, cap)] = q16[:cap]
        out = torch.empty((W * (cap + 1), d), dtype=torch.float16, device=q16.device)
        dist.all_gather_into_tensor(out, block)
        out = out.view(W, cap + 1, d)
        sizes = out[:, 0].contiguous().view(torch.int32)[:, 0].tolist()          # the one host sync
        if max(sizes) <= cap:
            break
        _QUERY_CAP = (max(sizes) + 63) // 64 * 64                                    # same decision on every rank: same headers
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


def _collective_device() -> torch.device:
    """where tensors handed to a collective must live: the GPU for nccl (= RCCL), the host for gloo"""
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def exchange_objects(per_dst: list) -> list:
    """Personalised all-to-all of python objects: rank r receives [per_dst[r] of rank 0, ..., of rank W-1]. Two collectives
    (sizes, then the pickled bytes with uneven splits): every rank gets exactly what was addressed to it, nothing else.
    ATLAS_EXCHANGE=allgather swaps it for one `all_gather_object` of the whole outbox (W x the bytes, the plainest collective
    there is): the switch to throw if a backend mishandles uneven or empty all-to-all splits."""
    if not is_initialized():
        return [per_dst[0]]
    import os
    import pickle

    if os.environ.get("ATLAS_EXCHANGE", "") == "allgather":
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
