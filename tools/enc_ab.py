"""Same-process A/B of the refresh encoder's GEMM configurations (dev tool, tuning build): the 512 x 128-token fp16 batch and the ragged
64..200 batch, configurations alternated round by round (boxes differ by several per cent: only same-process ratios count).
    python tools/enc_ab.py [cfgs, default 4,9] [rounds, default 6]
prints ms per batch per configuration (median, min) and, with diag bit 1, the same with the GEMM epilogues switched off."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import sys, time
import numpy as np
import torch
from atlas_amd import retrievers

cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "4,9").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
diags = [int(d) for d in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
m = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
NB = 512


def batch(lens, L_):
    ids = torch.randint(1000, 30522, (NB, L_), generator=g)
    mask = (torch.arange(L_)[None, :] < lens[:, None]).long()
    return (ids * mask).cuda(), mask.cuda()


work = {"full 512x128": batch(torch.full((NB,), 128), 128)}
lens = torch.randint(64, 201, (NB,), generator=g)
work["ragged 64..200"] = batch(lens, int(lens.max()))
out = torch.empty((NB, 768), dtype=torch.float16, device="cuda")
ref = {}
for name, (ids, mask) in work.items():
    for diag in diags:
        res = {c: [] for c in cfgs}
        for r in range(rounds):
            for c in cfgs:
                L.atlas_tune_set_gemm_cfg(c)
                L.atlas_tune_set_gemm_diag(diag)
                m.embed_into(out, ids, mask)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(4):
                    m.embed_into(out, ids, mask)
                torch.cuda.synchronize()
                res[c].append((time.perf_counter() - t) / 4 * 1e3)
                if diag == 0:
                    if (name, "out") not in ref:
                        ref[(name, "out")] = out.clone()
                    elif not torch.equal(out, ref[(name, "out")]):
                        print(f"!! cfg {c} differs from cfg {cfgs[0]} on {name}: max |d| = {(out.float() - ref[(name, 'out')].float()).abs().max().item():.3e}", flush=True)
        line = "  ".join(f"cfg {c}: {np.median(v):7.3f} ms (min {min(v):7.3f})" for c, v in res.items())
        print(f"{name:16s} diag {diag}: {line}   passages/s at median: " + " ".join(f"{NB / np.median(v) * 1e3:8.0f}" for v in res.values()), flush=True)
L.atlas_tune_set_gemm_cfg(-1)
L.atlas_tune_set_gemm_diag(0)
