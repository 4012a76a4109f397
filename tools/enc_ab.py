"""Same-process A/B of the refresh encoder's GEMM configurations (dev tool, tuning build): the 512 x 128-token fp16 batch and the ragged
64..200 batch, configurations alternated round by round (boxes differ by several per cent: only same-process ratios count).
    python tools/enc_ab.py [variants, default 4:0,9:0,4:1,9:1] [rounds, default 6]
a variant is cfg:diag (diag = the tuning hook's bits: 1 no epilogue, 4 nt stores, 8 sc1 stores, u << 8 start stagger in quarter-us steps);
prints ms per batch per variant (median, min); diag-0 variants must all give the same bits."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import sys, time
import numpy as np
import torch
from atlas_amd import retrievers

variants = [tuple(int(x) for x in v.split(":")) for v in (sys.argv[1] if len(sys.argv) > 1 else "4:0,9:0,4:1,9:1").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
for name, (ids, mask) in work.items():
    res = {v: [] for v in variants}
    ref = None
    for r in range(rounds):
        for v in variants:
            L.atlas_tune_set_gemm_cfg(v[0])
            L.atlas_tune_set_gemm_diag(v[1])
            m.embed_into(out, ids, mask)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(4):
                m.embed_into(out, ids, mask)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t) / 4 * 1e3)
            if not (v[1] & 1):
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(out, ref):
                    print(f"!! variant {v} differs from {variants[0]} on {name}: max |d| = {(out.float() - ref.float()).abs().max().item():.3e}", flush=True)
    for v, t in res.items():
        print(f"{name:16s} cfg {v[0]} diag {v[1]:5d}: {np.median(t):7.3f} ms (min {min(t):7.3f})  {NB / np.median(t) * 1e3:8.0f} passages/s", flush=True)
L.atlas_tune_set_gemm_cfg(-1)
L.atlas_tune_set_gemm_diag(0)
