"""Is the DMA-staged scan power-limited, and does it stay where it starts? (tuning build) 32M rows, back-to-back searches for SECS seconds per leg, rocm-smi
socket power / shader clock beside them, throughput per second of the leg: scan_kernel.h (dma 0) | dscan_kernel.h (dma 1) | dscan on an ALL-ZERO slab
(same instruction stream, no data toggling).
    python tools/scan_dma_power.py [rows] [secs]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import ctypes, re, subprocess, sys, threading, time
import numpy as np
import torch
from atlas_amd import HipDistributedIndex, _lib
from scan_policy_common import shard

L.atlas_tune_set_scan_dma.argtypes, L.atlas_tune_set_scan_dma.restype = [ctypes.c_int], None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
B, k, D = 64, 40, 768
slab = shard(N)
q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(int(L.atlas_scan_topk_workspace_bytes(N, B, D, k)), dtype=torch.uint8, device="cuda")


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True); self.stop = False; self.power = []; self.sclk = []; self.temp = []
    def run(self):
        while not self.stop:
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                break
            m = re.search(r"Power \(W\): ([\d.]+)", o); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
            t = re.findall(r"Temperature \(Sensor (\w+)\) \(C\): ([\d.]+)", o)
            if m: self.power.append(float(m.group(1)))
            if s: self.sclk.append(float(s.group(1)))
            if t: self.temp.append({a: float(b) for a, b in t})


def leg(name, mode, the_slab, the_q):
    L.atlas_tune_set_scan_dma(mode)
    def call():
        assert L.atlas_scan_topk_flags(the_q.data_ptr(), _lib.DT_F32, the_slab.data_ptr(), N, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                       ws.data_ptr(), ws.numel(), stream, None, None, _lib.SCAN_TRUST_PMAX) == 0
    for _ in range(5): call()
    torch.cuda.synchronize()
    smi = Smi(); smi.start()
    per_s, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < SECS:
        t1 = time.perf_counter(); n = 0
        while time.perf_counter() - t1 < 1.0:
            for _ in range(10): call()
            torch.cuda.synchronize(); n += 10
        per_s.append((time.perf_counter() - t1) / n * 1e3)
    smi.stop = True; smi.join(timeout=6)
    pw = np.array(smi.power[1:] or [0]); sc = np.array(smi.sclk[1:] or [0])
    temps = smi.temp[-1] if smi.temp else {}
    print(f"{name:44s}: ms per search by second of the leg {' '.join('%.3f' % x for x in per_s)}  ({N * 1536 / np.mean(per_s) / 1e9 / 8:.3f} of 8 TB/s)   power mean {pw.mean():.0f} W max {pw.max():.0f} W   "
          f"sclk mean {sc.mean():.0f} MHz min {sc.min():.0f}   temps at the end {temps}", flush=True)


for rnd in range(2):
    leg("scan_kernel<16,1,8> (dma 0)", 0, slab, q)
    leg("dscan_kernel<nt> (dma 1)", 1, slab, q)
zero = torch.zeros_like(slab)
leg("dscan_kernel<nt> on an all-zero slab, zero q", 1, zero, torch.zeros_like(q))
leg("dscan_kernel<nt> (dma 1), again", 1, slab, q)
L.atlas_tune_set_scan_dma(1)
