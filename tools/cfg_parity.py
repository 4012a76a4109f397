"""bit-equality of an experimental GEMM configuration with the default one (dev tool): 2-layer fp16 / bf16 encoder, ragged batch"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from atlas_amd import retrievers
cfgs = sys.argv[1:] or ["6"]
for dtype in (torch.float16, torch.bfloat16):
    torch.manual_seed(5)
    m = retrievers.Contriever(retrievers.BertConfigLite(num_hidden_layers=2)).to(dtype).eval().cuda().requires_grad_(False)
    g = torch.Generator().manual_seed(31)
    for n, L, lo in ((150, 128, 20), (512, 128, 128), (64, 200, 64)):
        lens = torch.randint(lo, L + 1, (n,), generator=g)
        ids = torch.randint(1000, 30522, (n, L), generator=g).cuda()
        mask = (torch.arange(L)[None, :] < lens[:, None]).long().cuda()
        L.atlas_tune_set_gemm_cfg(-1)
        L.atlas_tune_set_gemm_cfg(4)
        base = m(ids, mask)
        for c in cfgs:
            L.atlas_tune_set_gemm_cfg(int(c))
            got = m(ids, mask)
            print(dtype, (n, L, lo), "cfg", c, "identical" if torch.equal(got, base) else "DIFFERS max|d| = %g" % float((got.float() - base.float()).abs().max()), flush=True)
L.atlas_tune_set_gemm_cfg(-1)
