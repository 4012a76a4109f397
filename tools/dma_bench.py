import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "tools", "libdma_bench.so")
import torch
L = ctypes.CDLL(so)
L.dma_bench.restype = ctypes.c_float
L.dma_bench.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(16, dtype=torch.int32, device="cuda"); cyc = torch.zeros(2, dtype=torch.int64, device="cuda")
for kb in (1536, 6144):
    src = torch.randint(0, 255, (512 * kb,), dtype=torch.uint8, device="cuda")
    tiles, reps = kb // 128, 20
    for blocks in (256, 1):
        for mode in (0, 1, 2, 3):
            ms = L.dma_bench(mode, src.data_ptr(), kb, tiles, reps, blocks, out.data_ptr(), cyc.data_ptr())
            c = int(cyc.cpu()[0]); nbytes = tiles * reps * 65536
            print(f"rowbytes {kb} blocks {blocks:3d} mode {mode}: {ms*1e3:8.1f} us  {nbytes/c:6.1f} B/clk/CU (block 0)  {blocks*nbytes/ms/1e9:8.2f} TB/s aggregate  cycles/tile {c/(tiles*reps):7.0f}")
