"""The peer exchange (C-ABI atlas_xchg_*, dist_utils.PeerExchange): two PROCESSES sharing the one GPU of the box map each other's exchange
buffers through hipIpc handles and exchange packed winners through them -- the logic of the one-hop alternative to the all-gather of
src/index.py:134-151 (across two devices it has never run: no multi-GPU box). Every wait is bounded: a late peer is a status bit, not a hang."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> str:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.environ["ATLAS_ROOT"])
    import numpy as np, torch, torch.distributed as dist
    from atlas_amd import dist_utils
    from atlas_amd.index import merge_packed_host
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, W = dist.get_rank(), dist.get_world_size()
    B, k = 64, 40
    try:
        px = dist_utils.PeerExchange(slot_entries=B * k, wait_ms=2000)
    except Exception as e:                                   # hipIpc between two processes on one device not available here
        print("SKIP", repr(e)); sys.exit(0)
    rng = [np.random.default_rng(100 + r) for r in range(W)]
    for it in range(6):
        parts = [g.integers(1, 2**62, size=(B, k), dtype=np.int64) for g in rng]             # every rank knows every rank's winners
        if it == 3:
            parts = [np.sort(p, axis=1)[:, ::-1].copy() for p in parts]
            parts[0][:, 5:] = 0                                                              # padding entries
        out = px.exchange(torch.from_numpy(parts[rank]).cuda(), k)
        assert out is not None, "a peer was late"
        want = merge_packed_host(np.stack(parts), k)
        assert np.array_equal(out.cpu().numpy(), want), f"rank {rank} iteration {it}"
    # a late peer: rank 1 skips its push for one tag; rank 0 must report it within the bound instead of hanging
    px.wait_ms = 150
    if rank == 0:
        out = px.exchange(torch.from_numpy(parts[0]).cuda(), k)
        assert out is None
    else:
        px.tag += 1
    dist.barrier()
    px.close()
    print("OK", rank)
''')


@pytest.mark.gpu
def test_two_processes_exchange_packed_winners_through_mapped_buffers(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, ATLAS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("peer exchange worker timed out")
    if any("SKIP" in o for o in outs):
        pytest.skip("hipIpc mapping between two processes on one device is not available: " + outs[0][-300:])
    assert all(p.returncode == 0 for p in procs) and all("OK" in o for o in outs), "\n".join(o[-1500:] for o in outs)


KNN_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.environ["ATLAS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["ATLAS_ROOT"], "tests"))
    import numpy as np, torch, torch.distributed as dist
    import synth
    from oracle import oracle
    from atlas_amd import HipDistributedIndex
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, W = dist.get_rank(), dist.get_world_size()
    N, k, batches = 20_003, 12, (3, 5)
    P = synth.passages_f16(N, 768, 61); Qall = synth.queries_f32(sum(batches), 768, 62)
    lo = sum(batches[:rank]); Q = torch.from_numpy(Qall[lo: lo + batches[rank]]).cuda()
    mine = np.arange(rank, N, W)
    idx = HipDistributedIndex(exchange="peer")
    idx.init_embeddings([{"id": str(int(g)), "text": f"p{g}"} for g in mine])
    idx.embeddings[:, :] = torch.from_numpy(P[mine]).cuda().T
    for _ in range(3):
        docs, scores = idx.search_knn(Q, k)
    if idx.exchange != "peer":
        print("SKIP the peer exchange could not be set up"); sys.exit(0)
    s, i = oracle.search(oracle.f32_to_f16(Qall), P, k)
    ids = np.array([[int(d["id"]) for d in row] for row in docs], dtype=np.int64).reshape(len(docs), k)
    assert np.array_equal(ids, i[lo: lo + batches[rank]]) and np.array_equal(np.array(scores, dtype=np.float32), s[lo: lo + batches[rank]].astype(np.float32))
    dist.barrier()
    print("OK", rank)
''')


@pytest.mark.gpu
def test_search_knn_with_the_peer_exchange_equals_the_union(tmp_path):
    """HipDistributedIndex(exchange="peer") end to end with two processes on the one GPU (gloo for the host collectives): documents and
    scores of every rank equal the canonical search over the union of the shards"""
    script = tmp_path / "knn_worker.py"
    script.write_text(KNN_WORKER)
    env = dict(os.environ, ATLAS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("search_knn worker timed out")
    if any("SKIP" in o for o in outs):
        pytest.skip("peer exchange not available on this box")
    assert all(p.returncode == 0 for p in procs) and all("OK" in o for o in outs), "\n".join(o[-1500:] for o in outs)
