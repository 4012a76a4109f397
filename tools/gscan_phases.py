"""Shader-clock stamps of workgroup 0 of the GEMM-shaped scan (tuning build): what a k-tile's two phases are made of.
    python tools/gscan_phases.py [rows] [queries] [--certify]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys
import torch
from atlas_amd import _lib

FLAGS = _lib.SCAN_TRUST_PMAX
if "--certify" in sys.argv:      # the twin that measures every row norm (gscan_kernel<2, .>)
    sys.argv.remove("--certify"); FLAGS = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k, D = 40, 768
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((N, D), dtype=torch.float16, device="cuda")
for r0 in range(0, N, 1_000_000):
    n = min(1_000_000, N - r0); x = torch.randn((n, D), generator=g, device="cuda"); slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda").half()
out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
ws = torch.zeros(L.atlas_scan_topk_workspace_bytes(N, B, D, k), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
IT = 32
dbg = torch.zeros(8 * IT * 8 + 8 * 8 * 4 + 4, dtype=torch.int64, device="cuda")
def call():
    rc = L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F16, slab.data_ptr(), N, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                 ws.data_ptr(), ws.numel(), stream, None, None, FLAGS)
    assert rc == 0, rc
for _ in range(3): call()
torch.cuda.synchronize()
L.atlas_tune_set_scan_stamps(dbg.data_ptr())
call(); torch.cuda.synchronize()
L.atlas_tune_set_scan_stamps(None)
t = dbg.cpu()[:8 * IT * 8].view(8, IT, 8)
e = dbg.cpu()[8 * IT * 8:8 * IT * 8 + 8 * 8 * 4].view(8, 8, 4)
ck = dbg.cpu()[8 * IT * 8 + 8 * 8 * 4:]
if int(ck[2]) > int(ck[0]):
    print("workgroup 0 of the last scan launch: %d stamp cycles in %.1f us of the 100 MHz clock = %.3f GHz" % (int(ck[3] - ck[1]), (int(ck[2] - ck[0])) / 100.0, (int(ck[3] - ck[1])) / ((int(ck[2] - ck[0])) * 10.0)))
t0 = int(t[:, 0, 0].min())
names = ["reads", "dmaA/waitB", "bar1", "dmaB", "mfma", "waitA", "bar2"]
for w in (0, 1, 4, 5):
    print("wave", w, "(group %s)" % ("A" if w < 4 else "B"))
    for it in range(0, IT - 1):
        r = t[w, it]
        print("  it %2d (kt %2d) start %7d  " % (it + 24, (it + 24) % 12, int(r[0]) - t0) + "  ".join("%s %5d" % (names[i], int(r[i + 1] - r[i])) for i in range(7)) + "   rest %5d  iter %5d" % (int(t[w, it + 1, 0] - r[7]), int(t[w, it + 1, 0] - r[0])))
per = (t[:, IT - 1, 0] - t[:, 0, 0]).float() / (IT - 1)
print("cycles per k-tile by wave:", [int(x) for x in per.tolist()])
for w in (0, 4):
    print("epilogue of wave", w, "tiles 0..7: flags / dispatch cycles:", [(int(e[w, i, 1] - e[w, i, 0]), int(e[w, i, 2] - e[w, i, 1]), "hit fragments %d, buffer fill %d" % (int(e[w, i, 3]) >> 32, int(e[w, i, 3]) & 0xffffffff)) for i in range(8)])
print("candidates per query:", int(out_st[_lib.ST_N_CANDIDATES]) / B)
