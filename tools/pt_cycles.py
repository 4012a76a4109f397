"""Shader cycles and wall time of workgroup 0 of the persistent refresh GEMM (gemm_pt_kernel, tuning build) from the two stamps at its ends only
(an UNPERTURBED kernel: no stamp inside the counted vmcnt stream), per GEMM of a layer and per `diag` mode (tools/gemm_diag.py; 1 no epilogue,
64 no MFMAs, 128 no LDS-DMA pieces, 16 / 32 operand loads aliased to the first tile). A production pass runs first: the modes that skip the epilogue
leave the real activations in place.   python tools/pt_cycles.py [modes, default 0,1,65,129,193,49]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys
import torch
from atlas_amd import retrievers

L.atlas_tune_set_gemm_cfg(9)
m = retrievers.Contriever(retrievers.BertConfigLite(num_hidden_layers=1)).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (512, 128), generator=g).cuda()
mask = torch.ones((512, 128), dtype=torch.int64).cuda()
out = torch.empty((512, 768), dtype=torch.float16, device="cuda")
for _ in range(3): m.embed_into(out, ids, mask)
torch.cuda.synchronize()
WARM = 40
kinds = {"qkv": (1, 9 * 12), "out": (2, 3 * 12), "ffn1": (3, 12 * 12), "ffn2": (4, 3 * 48)}      # nth gemm_pt launch of the pass, k-tiles of workgroup 0
modes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,65,129,193,49").split(",")]
for md in modes:
    L.atlas_tune_set_gemm_diag(md)
    for _ in range(2): m.embed_into(out, ids, mask)          # the clock settles on this mode's load
    line = []
    for kind, (nth, nkt) in kinds.items():
        best = None
        for rep in range(3):
            dbg = torch.zeros(2048 + 8 * 32 * 8, dtype=torch.int64, device="cuda")
            L.atlas_tune_set_gemm_stamps_nth(dbg.data_ptr() | 1, 4 * WARM + nth)     # the stamped launch sits behind WARM back-to-back passes: sustained clocks
            for _ in range(WARM + 1): m.embed_into(out, ids, mask)
            torch.cuda.synchronize()
            L.atlas_tune_set_gemm_stamps_nth(None, 0)
            ck = dbg[1024:1028].cpu()
            us, cyc = int(ck[2] - ck[0]) / 100.0, int(ck[3] - ck[1])
            if best is None or us < best[0]: best = (us, cyc)
        us, cyc = best
        line.append("%s %6.1f us %7d cyc %.2f GHz %5d cyc/k-tile" % (kind, us, cyc, cyc / (us * 1e3), cyc // nkt))
    print("diag %3d: " % md + " | ".join(line), flush=True)
L.atlas_tune_set_gemm_diag(0)
