"""100 MHz stamps of workgroup 0 of the persistent GEMM around its tile boundaries (dev tool, tuning build): where does a tile-round's
time outside the k-loop go? One 1-layer encoder pass per GEMM kind; the stamped launch is the LAST gemm_pt launch of the pass, so the
layer is cut short after the GEMM of interest with ATLAS_PT_STAMP = qk | v | out | ffn1 | ffn2 (how many GEMMs are let through).
    python tools/pt_stamps.py"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import ctypes
import torch
from atlas_amd import retrievers

L.atlas_tune_set_gemm_cfg(9)
m = retrievers.Contriever(retrievers.BertConfigLite(num_hidden_layers=1)).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (512, 128), generator=g).cuda()
mask = torch.ones((512, 128), dtype=torch.int64).cuda()
out = torch.empty((512, 768), dtype=torch.float16, device="cuda")
m.embed_into(out, ids, mask); torch.cuda.synchronize()
names = ["tile start", "LAST: reads done", "LAST: barrier 1 passed", "LAST: MFMAs issued", "LAST: own wait done", "LAST: barrier 2 passed",
         "pre-epilogue wait done", "epilogue issued", "acc zeroed", "first iteration: barrier 1", "first iteration: MFMAs issued", "first iteration: end"]
for which, nth in (("ffn2 (EPI 2, K = 3072)", 5), ("ffn1 (EPI 1)", 4), ("out-proj (EPI 2)", 3), ("v (EPI 4)", 2), ("qk (EPI 3)", 1)):
    dbg = torch.zeros(8 * 8 * 16 + 16, dtype=torch.int64, device="cuda")
    L.atlas_tune_set_gemm_stamps_nth(dbg.data_ptr(), nth)
    m.embed_into(out, ids, mask); torch.cuda.synchronize()
    L.atlas_tune_set_gemm_stamps_nth(None, 0)
    t = dbg.cpu()[:1024].view(8, 8, 16)
    ck = dbg.cpu()[1024:1028]
    print("==", which, "-- workgroup 0: %.1f us, shader clock %.3f GHz" % (int(ck[2] - ck[0]) / 100.0, int(ck[3] - ck[1]) / (int(ck[2] - ck[0]) * 10.0)) if int(ck[2]) > int(ck[0]) else "")
    for w in (0, 4):
        for ti in (1, 2):
            r = t[w, ti]
            if int(r[0]) == 0:
                continue
            base = int(r[1])
            seq = [(names[i], int(r[i]) - base) for i in (1, 2, 3, 4, 5, 6, 7, 8)]
            nxt = t[w, ti + 1]
            seq += [(names[i], int(nxt[i]) - base) for i in (9, 10, 11) if int(nxt[i])]
            print(f"  wave {w} tile {ti}: k-loop {(int(r[1]) - int(r[0])) / 100:.2f} us; from the LAST iteration's reads, in us: " + "  ".join(f"{n} {v / 100:.2f}" for n, v in seq))
