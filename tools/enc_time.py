"""time the encoder on a few batch shapes (dev tool): full-length passages, ragged passages, query-like padding"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlas_amd import retrievers

dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
trim = len(sys.argv) > 2 and sys.argv[2] == "trim"
m = retrievers.Contriever(retrievers.BertConfigLite()).to(dtype).eval().cuda().requires_grad_(False)
print("dtype", dtype, "trim", trim)
g = torch.Generator().manual_seed(1)


def run(name, n, L, lens):
    ids = torch.randint(1000, 30522, (n, L), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids, mask = ids.cuda(), mask.cuda()
    out = torch.empty((n, 768), dtype=dtype, device="cuda")
    for _ in range(2):
        m.embed_into(out, ids, mask, trim_padding=trim)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        m.embed_into(out, ids, mask, trim_padding=trim)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    real = int(lens.clamp(max=L).sum())
    print(f"{name:28s} n={n} L={L} real tokens={real:7d}  {dt*1e3:8.2f} ms  {n/dt:9.0f} rows/s  {real/dt/1e6:6.2f} Mtok/s", flush=True)


run("passages full", 512, 128, torch.full((512,), 128))
run("passages ragged 64..128", 512, 128, torch.randint(64, 129, (512,), generator=g))
run("passages full L=512", 128, 512, torch.full((128,), 512))
run("queries 64 x pad512 (~20)", 64, 512, torch.randint(8, 33, (64,), generator=g))
run("queries 64 x L=32", 64, 32, torch.randint(8, 33, (64,), generator=g))
run("queries 8 x pad512 (~20)", 8, 512, torch.randint(8, 33, (8,), generator=g))
