import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "tools", "libdma_bench.so")
import torch
L = ctypes.CDLL(so)
L.dma_bench.restype = ctypes.c_float
L.dma_bench.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
out = torch.zeros(16, dtype=torch.int32, device="cuda"); cyc = torch.zeros(2, dtype=torch.int64, device="cuda")
for kb in (1536, 6144):
    src = torch.randint(0, 255, (258 * 256 * kb,), dtype=torch.uint8, device="cuda")     # W rows + 257 activation tiles
    tiles, reps = kb // 128, 1
    for share in (0, 12, 3, 1):
      for deep in (0, 1):
        for mode in (0,):
            ms = L.dma_bench(mode, src.data_ptr(), kb, tiles, reps, 256, out.data_ptr(), cyc.data_ptr(), share, deep)
            c = int(cyc.cpu()[0]); nbytes = tiles * reps * 65536
            what = {0: "all blocks the same 512 rows", 12: "activation tile shared by 12 blocks", 3: "shared by 3 blocks", 1: "distinct per block",
                    -12: "BLOCKED layout, shared by 12", -3: "BLOCKED layout, shared by 3", -1: "BLOCKED layout, distinct"}[share]
            print(f"deep={deep} rowbytes {kb} mode {mode} ({'LDS-DMA' if mode == 0 else 'to registers'}) {what:38s}: {nbytes/c:6.1f} B/clk/CU (block 0)  {256*nbytes/ms/1e9:7.2f} TB/s aggregate  cycles/k-tile {c/(tiles*reps):7.0f}")
