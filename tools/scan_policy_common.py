"""shared by the scan A/B tools: the benchmark's shard (L2-normalised gaussian rows, fp16), built on the device"""
import torch


def shard(rows, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    slab = torch.empty((rows, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, rows, 250_000):
        n = min(250_000, rows - r0)
        x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    return slab
