"""PMC workload: two full 512 x 128 batches through the fp16 encoder (dev tool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from atlas_amd import retrievers
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = retrievers.Contriever(retrievers.BertConfigLite(num_hidden_layers=layers)).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (512, 128), generator=g).cuda()
mask = torch.ones((512, 128), dtype=torch.int64).cuda()
out = torch.empty((512, 768), dtype=torch.float16, device="cuda")
for _ in range(2):
    m.embed_into(out, ids, mask)
torch.cuda.synchronize()
print("done")
