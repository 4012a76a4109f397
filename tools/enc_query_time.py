"""Query embedding as src/atlas.py:90-104 issues it (round 6): 64 queries of ~20 real tokens, tokenised with padding='max_length' (atlas.py:184-191:
padded to min(text_maxlength, 512) tokens), model precision fp32 | bf16 | fp16 -- `Contriever.forward` wall time per call (synchronised), for padded
widths 32 / 200 / 512, with and without trim_padding.   python tools/enc_query_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atlas_amd import retrievers

g = torch.Generator().manual_seed(1)
lens = torch.randint(8, 33, (64,), generator=g)
for name, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
    torch.manual_seed(0)
    m = retrievers.Contriever(retrievers.BertConfigLite()).to(dtype).eval().cuda().requires_grad_(False)
    for width in (32, 200, 512):
        ids = torch.randint(1000, 30522, (64, width), generator=g)
        mask = (torch.arange(width)[None, :] < lens[:, None]).long()
        ids, mask = (ids * mask).cuda(), mask.cuda()
        for trim in (False, True):
            for graphs in (False, True):
                m.query_graphs = graphs
                with torch.no_grad():
                    for _ in range(5):
                        e = m(ids, mask, trim_padding=trim)
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(30):
                        t = time.perf_counter(); e = m(ids, mask, trim_padding=trim); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
                print(f"{name} 64 queries x {float(lens.float().mean()):.0f} real tokens padded to {width:3d}, trim_padding={trim!s:5}, hipGraph replay={graphs!s:5}: "
                      f"{np.median(ts) * 1e3:7.3f} ms per call (min {min(ts) * 1e3:7.3f})", flush=True)
