"""Shader-cycle stamps of workgroup 0 of the persistent GEMM (gemm_pt_kernel, tuning build): what a k-tile's two phases are made of, iterations
24 .. 55 of the workgroup (its third to fifth tile at K = 768). One 1-layer encoder pass per GEMM kind; the stamped launch is the nth gemm_pt
launch of the pass.   python tools/pt_phases.py [kinds, default qk,ffn2]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys
import torch
from atlas_amd import retrievers

L.atlas_tune_set_gemm_cfg(9)
m = retrievers.Contriever(retrievers.BertConfigLite(num_hidden_layers=1)).half().eval().cuda().requires_grad_(False)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30522, (512, 128), generator=g).cuda()
mask = torch.ones((512, 128), dtype=torch.int64).cuda()
out = torch.empty((512, 768), dtype=torch.float16, device="cuda")
m.embed_into(out, ids, mask); torch.cuda.synchronize()
nth = {"qk": 1, "v": 2, "out": 3, "ffn1": 4, "ffn2": 5}
names = ["reads+pieces", "wait", "bar1", "-", "mfma", "waitA", "bar2"]
for kind in (sys.argv[1] if len(sys.argv) > 1 else "qk,ffn2").split(","):
    dbg = torch.zeros(2048 + 8 * 32 * 8, dtype=torch.int64, device="cuda")
    L.atlas_tune_set_gemm_stamps_nth(dbg.data_ptr(), nth[kind])
    m.embed_into(out, ids, mask); torch.cuda.synchronize()
    L.atlas_tune_set_gemm_stamps_nth(None, 0)
    d = dbg.cpu()
    ck = d[1024:1028]
    print("==", kind, "-- workgroup 0: %.1f us, shader clock %.3f GHz" % (int(ck[2] - ck[0]) / 100.0, int(ck[3] - ck[1]) / (int(ck[2] - ck[0]) * 10.0)))
    t = d[2048:].view(8, 32, 8)
    for w in (0, 4):
        print("  wave", w, "(group %s)" % ("A" if w < 4 else "B"))
        for it in range(0, 26):
            r = t[w, it]
            if int(r[0]) == 0:
                continue
            seg = [int(r[1] - r[0]), int(r[2] - r[1]), int(r[3] - r[2]), 0, int(r[5] - r[3]), int(r[6] - r[5]), int(r[7] - r[6])]
            print("    it %2d  " % (it + 24) + "  ".join("%s %5d" % (names[i], seg[i]) for i in (0, 1, 2, 4, 5, 6)) + "   rest %5d  iter %5d" % (int(t[w, it + 1, 0] - r[7]), int(t[w, it + 1, 0] - r[0])))
