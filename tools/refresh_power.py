"""Socket power and shader clock (rocm-smi) while the refresh encoder runs back to back: is that leg power-limited too?

    python tools/refresh_power.py
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import re, subprocess, threading, time
import numpy as np
import torch
from atlas_amd import retrievers


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True); self.stop = False; self.power = []; self.sclk = []
    def run(self):
        while not self.stop:
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                break
            m = re.search(r"Power \(W\): ([\d.]+)", o); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
            if m: self.power.append(float(m.group(1)))
            if s: self.sclk.append(float(s.group(1)))


enc = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (512, 128), generator=g).cuda()
mask = torch.ones((512, 128), dtype=torch.int64).cuda()
out = torch.empty((512, 768), dtype=torch.float16, device="cuda")
for _ in range(5): enc.embed_into(out, ids, mask)
torch.cuda.synchronize()
for rnd in range(2):
    smi = Smi(); smi.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(10): enc.embed_into(out, ids, mask)
        torch.cuda.synchronize(); n += 10
    dt = (time.perf_counter() - t0) / n
    smi.stop = True; smi.join(timeout=6)
    pw = np.array(smi.power[1:] or [0]); sc = np.array(smi.sclk[1:] or [0])
    print(f"refresh: {512 / dt:.0f} passages/s ({dt * 1e3:.2f} ms per 512 x 128-token batch)   power mean {pw.mean():.0f} W max {pw.max():.0f} W   sclk mean {sc.mean():.0f} MHz min {sc.min():.0f}   ({len(pw)} samples)", flush=True)
