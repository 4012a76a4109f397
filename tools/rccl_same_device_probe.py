"""Can RCCL run TWO ranks on ONE GPU here? (NCCL refuses duplicate devices; if RCCL does too, every multi-rank RCCL path stays untested on one-GPU boxes.)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 tools/rccl_same_device_probe.py"""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.full((4,), float(rank + 1), device="cuda")
    dist.all_reduce(t); torch.cuda.synchronize()
    out = torch.empty((world * 3,), device="cuda"); dist.all_gather_into_tensor(out, torch.full((3,), float(rank), device="cuda")); torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce -> {t.tolist()}, all_gather -> {out.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(3)
