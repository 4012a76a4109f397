"""Experiment (tuning build): the refresh batch as two half-batches on two plain streams (no CU masks), so that workgroups of DIFFERENT
kernels share a CU and one's GEMM epilogue meets the other's k-loop. Per GEMM configuration (4 = product: ping-pong, one workgroup
owns the CU; 6 = the two-workgroups-per-CU kernel for every GEMM): passages/s with one stream x 512 passages and with
two streams x 256, alternated.

    python tools/two_stream_refresh.py
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import ctypes, time
import torch
from atlas_amd import retrievers

NB, LEN = 512, 128
enc = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (NB, LEN), generator=g).cuda()
mask = torch.ones((NB, LEN), dtype=torch.int64).cuda()
out = torch.empty((NB, 768), dtype=torch.float16, device="cuda")
w = enc._pack()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(P, reps=12):
    nb = NB // P
    need = L.atlas_contriever_workspace_bytes(nb, LEN, w.dtype)
    wss = [torch.empty(int(need), dtype=torch.uint8, device="cuda") for _ in range(P)]
    o = torch.zeros_like(out)

    def one_pass():
        for p in range(P):
            a, b = p * nb, (p + 1) * nb
            rc = L.atlas_contriever_embed(ctypes.byref(w), ids[a:b].data_ptr(), mask[a:b].data_ptr(), None, nb, LEN, o[a:b].data_ptr(),
                                          wss[p].data_ptr(), wss[p].numel(), streams[p].cuda_stream)
            assert rc == 0, rc
    one_pass(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        one_pass()
    torch.cuda.synchronize()
    return NB * reps / (time.perf_counter() - t), o


ref = None
for rnd in range(3):
    for cfg in (-1, 6, 7):
        L.atlas_tune_set_gemm_cfg(cfg)
        r1, o1 = run(1)
        r2, o2 = run(2)
        if ref is None:
            ref = o1.clone()
        same = bool(torch.equal(o1, ref)) and bool(torch.equal(o2, ref))
        print(f"cfg {cfg:2d}: one stream x 512: {r1:8.0f} passages/s   two streams x 256: {r2:8.0f} passages/s ({r2 / r1:.3f} x)   identical={same}", flush=True)
L.atlas_tune_set_gemm_cfg(-1)
