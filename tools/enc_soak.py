"""Soak of two BUILDS of the library against each other (dev tool, round 5: the refresh GEMM's reads ahead): random batch shapes (512..1536 passages,
lengths uniform in [lo, hi] with random lo / hi up to 256, a few full-length ones), fp16 and bf16, every embedding of build B compared bit for bit
with build A's; a schedule race shows up as an intermittent mismatch.   python tools/enc_soak.py a=libA.so b=libB.so [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlas_amd import _lib, retrievers

libs, secs = [], 60.0
for a in sys.argv[1:]:
    if "=" in a:
        n, path = a.split("=", 1); libs.append((n, _lib._bind(os.path.abspath(path))))
    else:
        secs = float(a)
assert len(libs) == 2
g = torch.Generator().manual_seed(7)
models = {dt: retrievers.Contriever(retrievers.BertConfigLite(num_hidden_layers=3)).to(dt).eval().cuda().requires_grad_(False) for dt in (torch.float16, torch.bfloat16)}
t_end, n, bad = time.time() + secs, 0, 0
while time.time() < t_end:
    dt = (torch.float16, torch.bfloat16)[n & 1]
    m = models[dt]
    nb = int(torch.randint(512, 1537, (1,), generator=g))
    hi = int(torch.randint(32, 257, (1,), generator=g)); lo = int(torch.randint(1, hi + 1, (1,), generator=g))
    lens = torch.full((nb,), hi) if n % 7 == 0 else torch.randint(lo, hi + 1, (nb,), generator=g)
    L_ = int(lens.max())
    ids = torch.randint(1000, 30522, (nb, L_), generator=g)
    mask = (torch.arange(L_)[None, :] < lens[:, None]).long()
    ids, mask = (ids * mask).cuda(), mask.cuda()
    outs = []
    for name, h in libs:
        m._library = h
        out = torch.empty((nb, 768), dtype=dt, device="cuda")
        m.embed_into(out, ids, mask)
        outs.append(out)
    torch.cuda.synchronize()
    same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    finite = bool(torch.isfinite(outs[1].float()).all())
    if not same or not finite:
        bad += 1
        print(f"!! batch {n}: {dt} nb={nb} lens {lo}..{hi}: identical={same} finite={finite} max|d|={(outs[0].float() - outs[1].float()).abs().max().item():.3e}", flush=True)
    n += 1
print(f"{n} batches ({int(int(mask.sum()) > 0)}), {bad} mismatches between {libs[0][0]} and {libs[1][0]}", flush=True)
sys.exit(1 if bad else 0)
