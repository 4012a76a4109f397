"""Same-process A/B of the refresh encoder under knobs of the TUNING build (round 6), with energy per passage beside the time:
    python tools/enc_knob_ab.py [variants, default att0,att2,att3,noln] [seconds per leg, default 3] [rounds, default 3]
  att0 / att2 / att3   attention: one workgroup per item (the product) | the persistent prefetching kernel with 2 | 3 workgroups per CU
  att0x0               attention: one workgroup per item, item = blockIdx (rounds 1-5) instead of the XCD-aware mapping
  noln                 the two ln_kernel launches of every layer left out (RESULTS WRONG): the upper bound of ANY LayerNorm fusion, power effects included
Each leg runs the batch back to back for the given seconds beside a rocm-smi sampler: ms per batch, W, J per passage (= W x ms / 512). Under the
board's 1 400 W limit the time follows the energy, so J per passage is the quantity a schedule change has to move (VERDICT r05 next #3b).
Variants without `noln` must give identical bits."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys, time
import numpy as np
import torch
from atlas_amd import retrievers
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from bench import _SmiSampler

variants = (sys.argv[1] if len(sys.argv) > 1 else "att0,att2,att3,noln").split(",")
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
m = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
NB = 512


def batch(lens, L_):
    ids = torch.randint(1000, 30522, (NB, L_), generator=g)
    mask = (torch.arange(L_)[None, :] < lens[:, None]).long()
    return (ids * mask).cuda(), mask.cuda()


def select(v):
    L.atlas_tune_set_att_pf({"att0": 0, "att3": 3}.get(v, 2))
    L.atlas_tune_set_skip_ln(1 if v == "noln" else 0)
    L.atlas_tune_set_att_xmap(0 if v in ("att0x0", "att2", "att3") else 1)


work = {"full 512x128": batch(torch.full((NB,), 128), 128)}
lens = torch.randint(64, 129, (NB,), generator=g)
work["ragged 64..128"] = batch(lens, int(lens.max()))
out = torch.empty((NB, 768), dtype=torch.float16, device="cuda")
for name, (ids, mask) in work.items():
    res = {v: [] for v in variants}
    ref = None
    for r in range(rounds):
        for v in variants:
            select(v)
            for _ in range(3):
                m.embed_into(out, ids, mask)
            torch.cuda.synchronize()
            if v != "noln":
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(out, ref):
                    print(f"!! variant {v} differs from {variants[0]} on {name}: max |d| = {(out.float() - ref.float()).abs().max().item():.3e}", flush=True)
            sm = _SmiSampler(); sm.start()
            t, n = time.perf_counter(), 0
            while time.perf_counter() - t < seconds:
                for _ in range(10):
                    m.embed_into(out, ids, mask)
                torch.cuda.synchronize(); n += 10
            ms = (time.perf_counter() - t) / n * 1e3
            pw = sm.finish()
            res[v].append((ms, pw["watts_mean"] if pw else float("nan"), pw["sclk_mhz_mean"] if pw and pw["sclk_mhz_mean"] else float("nan")))
    for v, t in res.items():
        a = np.array(t)
        ms, w, clk = np.median(a[:, 0]), np.nanmedian(a[:, 1]), np.nanmedian(a[:, 2])
        print(f"{name:16s} {v:6s}: {ms:7.3f} ms per batch (min {a[:, 0].min():7.3f})  {NB / ms * 1e3:8.0f} passages/s  {w:6.0f} W  {clk:5.0f} MHz  {w * ms * 1e-3 / NB * 1e3:7.3f} mJ per passage", flush=True)
select("att2")
