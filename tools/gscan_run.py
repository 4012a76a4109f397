"""One (rows, queries) point of the big-batch search, product library, for rocprofv3 passes (kernel trace / PMC):
    python tools/gscan_run.py ROWS QUERIES [REPS] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlas_amd import _lib

N, B = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
k = int(sys.argv[4]) if len(sys.argv) > 4 else 40
D = 768
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((N, D), dtype=torch.float16, device="cuda")
for r0 in range(0, N, 1_000_000):
    n = min(1_000_000, N - r0); x = torch.randn((n, D), generator=g, device="cuda"); slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda").half()
out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
ws = torch.zeros(L.atlas_scan_topk_workspace_bytes(N, B, D, k), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    rc = L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F16, slab.data_ptr(), N, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                 ws.data_ptr(), ws.numel(), stream, None, None, _lib.SCAN_TRUST_PMAX)
    assert rc == 0, rc
torch.cuda.synchronize()
print(N, B, out_st[:8].tolist())
