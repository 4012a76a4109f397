"""cycle stamps of block 0 of the ping-pong GEMM (dev tool): one 2-layer encoder pass, stamps from the LAST GEMM launched"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from atlas_amd import retrievers, _lib
L.atlas_tune_set_gemm_cfg(int(os.environ.get("ATLAS_GEMM_CFG", "4")))
L = _lib.lib()
L.atlas_tune_set_gemm_stamps.argtypes = [ctypes.c_void_p]
m = retrievers.Contriever(retrievers.BertConfigLite(num_hidden_layers=1)).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (512, 128), generator=g).cuda()
mask = torch.ones((512, 128), dtype=torch.int64).cuda()
out = torch.empty((512, 768), dtype=torch.float16, device="cuda")
m.embed_into(out, ids, mask); torch.cuda.synchronize()
dbg = torch.zeros(8 * 16 * 8 + 16, dtype=torch.int64, device="cuda")
L.atlas_tune_set_gemm_stamps(dbg.data_ptr())
m.embed_into(out, ids, mask); torch.cuda.synchronize()        # last GEMM = FF2 (K = 3072, 48 k-tiles; first 16 stamped)
L.atlas_tune_set_gemm_stamps(None)
tail = dbg.cpu()[1024:1029].tolist()
sc, rc, nk = tail[2] - tail[0], tail[3] - tail[1], tail[4]
print("k-loop of block 0: %d k-tiles, %d shader cycles in %.2f us (100 MHz clock) -> %.0f MHz, %.0f cycles = %.3f us per k-tile (stamps on)" % (nk, sc, rc / 100.0, sc / rc * 100.0, sc / nk, rc / 100.0 / nk))
t = dbg.cpu()[:1024].view(8, 16, 8)
t0 = int(t[:, 0, 0].min())
names = ["-", "reads", "dmaA/vmwB", "bar1", "dmaB", "mfma", "vmwA", "bar2"][1:]
for w in (0, 1, 4, 5):
    print("wave", w)
    for kt in range(2, 8):
        r = t[w, kt]
        print("  kt %2d start %7d  " % (kt, int(r[0]) - t0) + "  ".join("%s %5d" % (names[i], int(r[i + 1] - r[i])) for i in range(7)) + "   iter %5d" % int(t[w, kt + 1, 0] - r[0]))
