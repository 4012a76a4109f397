"""`python bench.py --gpus N` -- the driver's plain command shape, no torch.distributed.run around it -- must launch its N ranks itself
(it used to die on `assert world == args.gpus` before a single kernel ran).

CPU box: the launcher and the rendezvous (ATLAS_BENCH_RENDEZVOUS_ONLY=1: the ranks meet over gloo and rank 0 says so), the loud refusal
without a GPU (every rank refuses, the launcher returns non-zero, no JSON line), the WORLD_SIZE / --gpus mismatch message.
GPU box (one GPU is enough): exactly that command shape with ATLAS_BENCH_BACKEND=gloo -- two ranks sharing the GPU, the all-gather staged
through the host -- prints ONE JSON line with n_gpus == 2 and the per-hop times; the same under torch.distributed.run.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _json_lines(out):
    return [ln for ln in out.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("n", [2, 3])
def test_plain_command_launches_its_own_ranks_and_they_meet(n):
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600,
                       env=_env(ATLAS_BENCH_RENDEZVOUS_ONLY="1"), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout                                     # rank 0 only
    assert json.loads(lines[0]) == {"rendezvous_ok": True, "world": n, "self_launched": True}


def test_plain_multi_gpu_command_refuses_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600,
                       env=_env(), cwd=ROOT)
    assert p.returncode != 0 and "MI355X" in p.stderr
    assert not _json_lines(p.stdout)                                      # no line from a fallback, from either rank


def test_world_size_mismatch_says_how_to_launch():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stderr and not _json_lines(p.stdout)


def _check_two_rank_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["passages_total"] == 1000003 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["passages_per_gpu"] == len(range(0, 1000003, 2)) and d["config"]["parallelism"] == "shard2+rccl-allgather"
    h = d["detail"]["hops"]
    assert h["all_gather_packed_ms"] > 0 and h["merge_packed_ms"] > 0 and h["scan_and_local_merge_ms"] > 0
    assert h["bytes_per_rank_all_gather"] == 64 * 40 * 8 and h["backend"] == "gloo"
    # value = queries x steps / max-over-ranks time
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    return d


@pytest.mark.gpu
def test_plain_two_rank_command_prints_one_line_on_one_gpu_over_gloo():
    """the command the driver types, with the two ranks sharing this box's GPU (gloo: a logic check of the N > 1 path, not a measurement)"""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--passages", "1000003", "--refresh-batches", "0", "--cpu-seconds", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(ATLAS_BENCH_BACKEND="gloo"), cwd=ROOT)
    d = _check_two_rank_line(p)
    # the N > 1 default: the exchange of step i on a second stream under the scan of step i + 1, with its serialised twin timed beside it
    assert d["detail"]["exchange_overlapped"] is True and d["detail"]["ms_per_step_serialized"] > 0
    p = subprocess.run(cmd + ["--overlap-exchange", "off"], capture_output=True, text=True, timeout=900, env=_env(ATLAS_BENCH_BACKEND="gloo"), cwd=ROOT)
    d = _check_two_rank_line(p)
    assert d["detail"]["exchange_overlapped"] is False and d["detail"]["ms_per_step_serialized"] is None


@pytest.mark.gpu
def test_two_ranks_under_torchrun_still_work_on_one_gpu_over_gloo():
    import socket

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--passages", "1000003",
                        "--refresh-batches", "0", "--cpu-seconds", "0", "--distinct-queries"], capture_output=True, text=True, timeout=900,
                       env=_env(ATLAS_BENCH_BACKEND="gloo"), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["distinct_queries"] is True and d["config"]["queries"] == 128
    assert d["detail"]["hops"]["bytes_per_rank_all_gather"] == 128 * 40 * 8
