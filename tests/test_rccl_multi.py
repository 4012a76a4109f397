"""The N > 1 path on REAL ranks: one process per GPU, `torch.distributed` backend `nccl` (= RCCL over xGMI). Skipped (cleanly) on
boxes with a single GPU -- the world-size-2/3 logic is covered on CPU by tests/test_dist_gloo.py; this file is what runs the day a
multi-GPU node is attached:

  * search_knn at W = 2, 4, 8 (those the node has) with uneven and empty per-rank batches, round-robin shards (src/index_io.py:41):
    every rank's documents and scores equal the canonical single-shard search over the union (CPU oracle), through the device-side
    pack -> ONE all_gather_into_tensor -> merge kernels and the personalised passage exchange over RCCL;
  * the device merge of the gathered candidates equals the host merge on every rank;
  * `bench.py --gpus N` under torch.distributed.run with a corpus size N does not divide: one JSON line, whole-job value.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _n_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, W, port, N, batches, k, out_dir, exchange="rccl"):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=W, device_id=dev)
    try:
        from atlas_amd import HipDistributedIndex, index as im

        P = synth.passages_f16(N, 768, 61)
        Qall = synth.queries_f32(sum(batches), 768, 62)
        lo = sum(batches[:rank])
        Q = torch.from_numpy(Qall[lo: lo + batches[rank]]).to(dev)
        mine = np.arange(rank, N, W)                                     # src/index_io.py:41
        idx = HipDistributedIndex(exchange=exchange)
        idx.init_embeddings([{"id": str(int(g)), "text": f"p{g}"} for g in mine])
        idx.embeddings[:, :] = torch.from_numpy(P[mine]).to(dev).T
        for _ in range(2):                                               # a collective: every rank calls it the same number of times
            docs, scores = idx.search_knn(Q, k)
        assert all(d["text"] == f"p{d['id']}" for row in docs for d in row)
        assert idx.exchange == exchange, "the peer exchange fell back to the collective"      # (reported, so that a silent fallback does not pass for coverage)
        ids = np.array([[int(d["id"]) for d in row] for row in docs], dtype=np.int64).reshape(len(docs), k)
        # device merge == host merge on what was actually gathered
        allq = torch.from_numpy(Qall).to(dev)
        s_d, r_d, s_h, r_h = idx._local_topk(allq, k)
        packed = idx._pack(s_d, r_d, s_h, r_h, W, rank)
        from atlas_amd import dist_utils

        gathered = dist_utils.all_gather_packed(packed)
        assert np.array_equal(idx._merge(gathered, k), im.merge_packed_host(gathered.cpu().numpy(), k))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids, scores=np.array(scores, dtype=np.float32).reshape(len(docs), k))
        dist.barrier()
    finally:
        dist.destroy_process_group()


# (both modes are plain tests: the peer exchange has never run across two devices -- its logic is tested with two processes on one GPU -- and the
#  day a node appears a failure there must turn the suite red, not pass as an expected one)
@pytest.mark.parametrize("exchange", ["rccl", "peer"])
@pytest.mark.parametrize("W,batches", [(2, (3, 5)), (2, (4, 0)), (4, (2, 0, 5, 1)), (8, (1, 2, 0, 3, 1, 0, 2, 4))])
def test_search_knn_over_rccl_equals_union(W, batches, exchange, tmp_path, oracle_mod):
    """both exchange modes: the all-gather of the packed winners, and the peer-mapped exchange buffers (index.py: exchange="peer")"""
    if _n_gpus() < W:
        pytest.skip(f"needs {W} GPUs, this box has {_n_gpus()}")
    import torch.multiprocessing as mp

    N, k = 50_003, 12                                                    # not a multiple of W: shards differ by one row
    mp.spawn(_worker, args=(W, _free_port(), N, batches, k, str(tmp_path), exchange), nprocs=W, join=True)
    P = synth.passages_f16(N, 768, 61)
    Q = synth.queries_f32(sum(batches), 768, 62)
    s, i = oracle_mod.search(oracle_mod.f32_to_f16(Q), P, k)
    lo = 0
    for r in range(W):
        got = np.load(os.path.join(tmp_path, f"r{r}.npz"))
        n = batches[r]
        if n:
            assert np.array_equal(got["ids"], i[lo: lo + n]), r
            assert np.array_equal(got["scores"], s[lo: lo + n].astype(np.float32)), r
        else:
            assert got["ids"].size == 0
        lo += n


@pytest.mark.parametrize("W", [2, 4, 8])
def test_bench_runs_under_torchrun(W):
    if _n_gpus() < W:
        pytest.skip(f"needs {W} GPUs, this box has {_n_gpus()}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={W}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--steps", "5", "--warmup", "2",
           "--passages", "1000003", "--refresh-batches", "0", "--cpu-seconds", "0"]
    # the metric step (replicated queries), the API step (every rank its own 64), and the metric step + the synchronous `search_knn` leg (opt-in at N > 1)
    for extra in ([], ["--distinct-queries"], ["--knn-leg"]):
        p = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]                         # rank 0 prints ONE line
        d = json.loads(lines[0])
        assert d["n_gpus"] == W and d["config"]["passages_total"] == 1000003 and d["value"] > 0
        assert d["config"]["passages_per_gpu"] == len(range(0, 1000003, W))
        distinct = "--distinct-queries" in extra
        assert d["config"]["distinct_queries"] is distinct and d["config"]["queries"] == (64 * W if distinct else 64)
        assert (d["detail"]["search_knn_ms_per_batch"] is not None) == ("--knn-leg" in extra)
        h = d["detail"]["hops"]
        assert h["all_gather_packed_ms"] > 0 and h["merge_packed_ms"] > 0 and h["bytes_per_rank_all_gather"] == d["config"]["queries"] * 40 * 8
        if distinct:
            assert d["roofline"]["bound"] == ("mfma" if 64 * W > 96 else "hbm")
