"""What a larger refresh batch is worth (round 6): the fp16 encoder on n x 128-token batches, n = 512 (the reference's per_gpu_embedder_batch_size
default, options.py:43-48) | 1024 | 2048 | 4096, alternated in one process: ms per 512 passages, W, mJ per passage. The streamed refresh forms
its own batches (atlas_amd.refresh.TOKEN_BUDGET tokens each; which passages share a batch changes no embedding), so a larger batch costs
nothing but workspace: every launch's fixed cost (ramp, first fetch from HBM, the tail of the last tiles: 7 launches per layer) is spread over
more tiles per CU.   python tools/enc_batch_size.py [sizes, default 512,1024,2048,4096] [seconds per leg, default 3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atlas_amd import retrievers
from bench import _SmiSampler

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048,4096").split(",")]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
torch.manual_seed(99)
m = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
L_ = 128
big = max(sizes)
ids = torch.randint(1000, 30522, (big, L_), generator=g).cuda()
mask = torch.ones((big, L_), dtype=torch.int64).cuda()
out = torch.empty((big, 768), dtype=torch.float16, device="cuda")
ref = None
res = {n: [] for n in sizes}
for rnd in range(3):
    for n in sizes:
        for _ in range(2):
            m.embed_into(out[:n], ids[:n], mask[:n])
        torch.cuda.synchronize()
        if ref is None:
            m.embed_into(out[:512], ids[:512], mask[:512]); torch.cuda.synchronize(); ref = out[:512].clone()
            m.embed_into(out[:n], ids[:n], mask[:n]); torch.cuda.synchronize()
        assert torch.equal(out[:512], ref), f"batch of {n}: the first 512 embeddings differ from the batch of 512"
        sm = _SmiSampler(); sm.start()
        t, k = time.perf_counter(), 0
        while time.perf_counter() - t < seconds:
            for _ in range(max(1, 5120 // n)):
                m.embed_into(out[:n], ids[:n], mask[:n])
            torch.cuda.synchronize(); k += max(1, 5120 // n)
        ms512 = (time.perf_counter() - t) / k * 1e3 * 512 / n
        pw = sm.finish()
        res[n].append((ms512, pw["watts_mean"] if pw else float("nan")))
for n, t in res.items():
    a = np.array(t)
    ms, w = np.median(a[:, 0]), np.nanmedian(a[:, 1])
    print(f"batch of {n:5d} x {L_}: {ms:7.3f} ms per 512 passages (min {a[:, 0].min():7.3f})  {512 / ms * 1e3:8.0f} passages/s  frac {512 / ms * 1e3 * 22351179776.0 / 2.5e15:.4f}  {w:6.0f} W  {w * ms * 1e-3 / 512 * 1e3:7.3f} mJ per passage", flush=True)
