"""Per-GEMM fixed cost vs k-loop cost of the 256x256 ping-pong kernel (dev tool, run under rocprofv3 --kernel-trace by scripts/gpu_gemm.sh):
the 512 x 128-token fp16 refresh batch with ATLAS_GEMM_DIAG modes 0 (production), 1 (no epilogue), 2 (no k-loop)."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from atlas_amd import retrievers, _lib
L = _lib.lib()
L.atlas_tune_set_gemm_diag.argtypes = [ctypes.c_int]
modes = [a for a in sys.argv[1:]] or ["0", "1", "2"]     # "diag" or "cfg:diag"
NB = int(os.environ.get("NB", "512"))
m = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (NB, 128), generator=g).cuda()
mask = torch.ones((NB, 128), dtype=torch.int64).cuda()
out = torch.empty((NB, 768), dtype=torch.float16, device="cuda")
for md in modes:
    if ":" in md:
        cfg_, md = md.split(":"); L.atlas_tune_set_gemm_cfg(int(cfg_))
    L.atlas_tune_set_gemm_diag(int(md))
    for _ in range(4):
        m.embed_into(out, ids, mask)
    torch.cuda.synchronize()
L.atlas_tune_set_gemm_diag(0)
