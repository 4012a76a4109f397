"""Same-process A/B of BUILDS for the CERTIFYING twin of the GEMM-shaped pass (gscan_kernel<2, FB>) next to the trusting one:
    python tools/gscan_cert_ab.py name=libA.so name=libB.so ... [rows, default 4000000] [--batches 128,256,512]
builds from `scripts/build_variant.sh <git rev> <out.so>`; rounds alternate the builds; outputs must be identical across builds and twins."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atlas_amd import _lib

D, k = 768, 40
libs, rows, batches = [], 4_000_000, (128, 256, 512)
args = sys.argv[1:]
if "--batches" in args:
    i = args.index("--batches"); batches = tuple(int(x) for x in args[i + 1].split(",")); del args[i:i + 2]
for a in args:
    if "=" in a:
        n, p = a.split("=", 1); libs.append((n, _lib._bind(os.path.abspath(p))))
    else:
        rows = int(a)
g = torch.Generator(device="cuda").manual_seed(1)
slab = torch.empty((rows, D), dtype=torch.float16, device="cuda")
for r0 in range(0, rows, 1_000_000):
    n = min(1_000_000, rows - r0); x = torch.randn((n, D), generator=g, device="cuda"); slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
stream = torch.cuda.current_stream().cuda_stream
for B in batches:
    q = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
    out_s = torch.empty((B, k), dtype=torch.float16, device="cuda"); out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    ws = {n: torch.zeros(int(L.atlas_scan_topk_workspace_bytes(rows, B, D, k)), dtype=torch.uint8, device="cuda") for n, L in libs}
    reps = max(3, int(200 * 64 / B))
    res, ref = {}, None
    for rnd in range(5):
        for n, L in libs:
            for tag, flags in (("trusting", _lib.SCAN_TRUST_PMAX), ("certifying", 0)):
                def call():
                    rc = L.atlas_scan_topk_flags(q.data_ptr(), _lib.DT_F32, slab.data_ptr(), rows, B, D, k, 1.002, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                 ws[n].data_ptr(), ws[n].numel(), stream, None, None, flags)
                    assert rc == 0, rc
                for _ in range(2): call()
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(reps): call()
                torch.cuda.synchronize()
                res.setdefault((n, tag), []).append((time.perf_counter() - t) / reps * 1e3)
                st = out_st[:8].tolist()
                assert st[_lib.ST_FLAGS] == 0, (n, tag, st)
                cur = (out_s.clone(), out_i.clone())
                if ref is None: ref = cur
                assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]), (n, tag, "results differ")
    line = f"{rows} rows x {B:4d} queries:"
    for n, _ in libs:
        t, c = float(np.median(res[(n, 'trusting')])), float(np.median(res[(n, 'certifying')]))
        line += f"  [{n}] trusting {t:6.3f} ms, certifying {c:6.3f} ms (+{(c / t - 1) * 100:4.1f} %)"
    print(line, flush=True)
