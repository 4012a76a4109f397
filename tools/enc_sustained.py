"""Sustained refresh rate (dev tool): the 512 x 128-token fp16 batch back to back for SECS seconds, throughput per window of 10 batches."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from atlas_amd import retrievers
enc = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (512, 128), generator=g).cuda()
mask = torch.ones((512, 128), dtype=torch.int64).cuda()
out = torch.empty((512, 768), dtype=torch.float16, device="cuda")
enc.embed_into(out, ids, mask); torch.cuda.synchronize()
t_end = time.time() + float(os.environ.get("SECS", "12"))
i = 0
while time.time() < t_end:
    t = time.perf_counter()
    for _ in range(10):
        enc.embed_into(out, ids, mask)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    if i % 5 == 0:
        smi = ""
        if i % 20 == 0:
            try:
                smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
                smi = " | ".join(l.strip() for l in smi.splitlines() if any(k in l for k in ("sclk", "Power (W)", "junction", "Socket Power", "Temperature (Sensor junction)")))[:300]
            except Exception as e:
                smi = repr(e)
        print(f"t={i * 10:5d} batches  {dt * 1e3:6.2f} ms/batch  {512 / dt:7.0f} passages/s  {smi}", flush=True)
    i += 1
