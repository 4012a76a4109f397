"""GPU tuning session: access-pattern streaming ceilings + scan_kernel variants (tools/microbench.hip)."""
import ctypes, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atlas_amd import HipDistributedIndex, _lib

mb = ctypes.CDLL(os.path.join(ROOT, "tools", "libatlas_mb.so"))
mb.mb_stream.restype = ctypes.c_float
mb.mb_stream.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
mb.mb_scan.restype = ctypes.c_float
mb.mb_scan.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_int, ctypes.c_int, ctypes.c_int]

def shard(rows, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    slab = torch.empty((rows, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, rows, 250_000):
        n = min(250_000, rows - r0)
        x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0+n] = (x / x.norm(dim=1, keepdim=True)).half()
    return slab

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000]
for N in sizes:
    slab = shard(N)
    q = torch.randn((64, 768), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
    idx = HipDistributedIndex(); idx._set_slab(slab)
    idx._compute_scores_and_indices(q, 40)                 # fills idx._ws with qfrag / qeps of these queries
    ws = idx._ws
    qfrag, qeps = ws.data_ptr(), ws.data_ptr() + 2 * 98304
    th_real = ws[2 * 98304 + 256: 2 * 98304 + 512].clone()            # theta0 of the product's sample pre-pass
    th_inf = torch.full((64,), float('-inf'), device='cuda')
    print('theta0[:4]', th_real.view(torch.float32)[:4].tolist())
    out = torch.zeros(1024, dtype=torch.int32, device="cuda")
    big = torch.empty(1024 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    gb = N * 1536 / 1e9
    res = {"N": N}
    for pat, ring in [(0, 8), (1, 4), (1, 8), (1, 24), (2, 4), (2, 12)]:
        ms = mb.mb_stream(pat, ring, slab.data_ptr(), N, out.data_ptr(), 20)
        res[f"stream_p{pat}_r{ring}"] = (round(ms, 4), round(gb / ms, 1))
    names = {0: "8,4,4", 1: "8,4,3", 2: "8,4,6", 3: "8,2,4", 4: "8,2,8", 5: "12,2,4", 6: "16,2,4", 7: "16,1,8", 8: "4,4,6"}
    for v, nm in names.items():
        for tag, nq, th in (("cold", 64, th_inf), ("theta0", 64, th_real), ("nq0", 0, th_inf)):
            ms = mb.mb_scan(v, slab.data_ptr(), N, qfrag, qeps, th.data_ptr(), big.data_ptr(), nq, 40, 20)
            res[f"scan<{nm}>_{tag}"] = (round(ms, 4), round(gb / ms, 1))
    print(json.dumps(res), flush=True)
    for k_, v_ in res.items():
        print(f"  {k_:24s} {v_}")
    del slab, big, idx
    torch.cuda.empty_cache()
