"""Is the scan power-limited? Per scan-kernel variant (tuning build): time per 32M-row scan over ~3 s of back-to-back calls, with the
socket power and the shader clock sampled by rocm-smi meanwhile.

    python tools/scan_power.py [rows] [variants...]
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import re, subprocess, sys, threading, time
import numpy as np
import torch
from atlas_amd import HipDistributedIndex, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
variants = [int(v) for v in sys.argv[2:]] or [0, 1, 3, 4, 2]
B, k, D = 64, 40, 768
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((N, D), dtype=torch.float16, device="cuda")
for r0 in range(0, N, 250_000):
    n = min(250_000, N - r0); x = torch.randn((n, D), generator=g, device="cuda")
    slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
q = torch.randn((B, D), generator=g, device="cuda")
out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


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


ref = None
names = {0: "<16,1,8>", 1: "<8,2,8>", 2: "<8,4,4>", 3: "<16,2,4>", 4: "<12,2,4>", 5: "<16,1,8,nt>", 6: "<16,1,8,sc1>"}
for rnd in range(2):
    for v in variants:
        L.atlas_tune_set_scan_variant(v)
        idx = HipDistributedIndex(); idx._set_slab(slab)
        s0, i0 = idx._compute_scores_and_indices(q, k)
        if ref is None: ref = (s0.clone(), i0.clone())
        ws, pmax = idx._ws, float(idx._pmax)
        def call():
            rc = L.atlas_scan_topk_ex(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(), ws.data_ptr(), ws.numel(), stream, None, None)
            assert rc == 0, rc
        for _ in range(20): call()
        torch.cuda.synchronize()
        smi = Smi(); smi.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 3.0:
            for _ in range(20): call()
            torch.cuda.synchronize(); n += 20
        dt = (time.perf_counter() - t0) / n * 1e3
        smi.stop = True; smi.join(timeout=6)
        ok = torch.equal(out_s, ref[0]) and torch.equal(out_i, ref[1])
        pw = np.array(smi.power[1:] or [0]); sc = np.array(smi.sclk[1:] or [0])
        print(f"variant {v} {names.get(v, '?'):12s}: {dt:.4f} ms per scan+merge ({N * 1536 / dt / 1e9 / 8:.3f} of 8 TB/s)   power mean {pw.mean():.0f} W max {pw.max():.0f} W   sclk mean {sc.mean():.0f} MHz   ({len(pw)} samples)  identical={ok}", flush=True)
        del idx
L.atlas_tune_set_scan_variant(0)
