"""Socket power and shader clock (rocm-smi) while the refresh encoder runs back to back: is that leg power-limited too?

    python tools/refresh_power.py [random zero random]

Round 5: the DATA is the knob. `zero` clears every parameter in place (the packed weight image follows the parameters' version counters), so
every GEMM operand, every activation and every stored tile is 0: the SAME instruction stream, launches and addresses with (almost) no bit
toggling in the matrix pipe, the LDS and the fabric. If the batch time drops with the power, the kernels are bound by the 1 400 W the board
gives them, not by their schedule.
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
saved = [q.detach().clone() for q in enc.parameters()]
for mode in (_sys.argv[1:] or ["random", "random"]):
    with torch.no_grad():
        for q, s in zip(enc.parameters(), saved):
            q.zero_() if mode == "zero" else q.copy_(s)
    for _ in range(3): enc.embed_into(out, ids, mask)
    torch.cuda.synchronize()
    assert mode != "zero" or float(out.float().abs().max()) == 0.0
    smi = Smi(); smi.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(10): enc.embed_into(out, ids, mask)
        torch.cuda.synchronize(); n += 10
    dt = (time.perf_counter() - t0) / n
    smi.stop = True; smi.join(timeout=6)
    pw = np.array(smi.power[1:] or [0]); sc = np.array(smi.sclk[1:] or [0])
    print(f"refresh [{mode:6s}]: {512 / dt:.0f} passages/s ({dt * 1e3:.2f} ms per 512 x 128-token batch)   power mean {pw.mean():.0f} W max {pw.max():.0f} W   sclk mean {sc.mean():.0f} MHz min {sc.min():.0f}   ({len(pw)} samples)", flush=True)
