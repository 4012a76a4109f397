"""one query-like batch (64 x ~20 tokens padded to 512) through the encoder, 20 times: for rocprofv3 --kernel-trace (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlas_amd import retrievers
dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
m = retrievers.Contriever(retrievers.BertConfigLite()).to(dtype).eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
lens = torch.randint(8, 33, (64,), generator=g)
ids = torch.randint(1000, 30522, (64, 32), generator=g).cuda()
mask = (torch.arange(32)[None, :] < lens[:, None]).long().cuda()
out = torch.empty((64, 768), dtype=dtype, device="cuda")
for _ in range(20):
    m.embed_into(out, ids, mask)
torch.cuda.synchronize()
