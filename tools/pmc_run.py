"""Workload for the PMC (FETCH_SIZE) pass: a streaming kernel with a KNOWN byte count in the scan's access
pattern (calibration: gfx950 FETCH_SIZE under-reports wide reads), then the product search."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
mb = ctypes.CDLL(os.path.join(ROOT, "tools", "libatlas_mb.so"))
mb.mb_stream.restype = ctypes.c_float
mb.mb_stream.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
for r0 in range(0, N, 250_000):
    n = min(250_000, N - r0); x = torch.randn((n, 768), generator=g, device="cuda")
    slab[r0:r0+n] = (x / x.norm(dim=1, keepdim=True)).half()
out = torch.zeros(1024, dtype=torch.int32, device="cuda")
print("stream p1 ms", mb.mb_stream(1, 8, slab.data_ptr(), N, out.data_ptr(), 3), "bytes", N * 1536)
print("stream p0 ms", mb.mb_stream(0, 8, slab.data_ptr(), N, out.data_ptr(), 3))
# (round 6) the calibration stream of dscan_kernel.h: LDS-DMA nt, 8 rows x 128 B per wave instruction, exactly rows x 1536 bytes as well
print("stream dma nt ms", mb.mb_stream(3, 8, slab.data_ptr(), N, out.data_ptr(), 3))
q = torch.randn((64, 768), device="cuda")
idx = HipDistributedIndex(); idx._set_slab(slab)
for _ in range(4):
    idx._compute_scores_and_indices(q, 40)
torch.cuda.synchronize()
print("done", idx.last_search_stats)
