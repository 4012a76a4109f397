"""Same-process A/B of BUILDS of the library (dev tool): boxes differ by several per cent and so do consecutive processes on one box, so two
source versions are compared by loading both shared objects into one process and alternating them round by round.
    python tools/lib_ab.py enc|scan|both  name=path/to/libA.so name=path/to/libB.so ...  [rounds]
enc : the refresh encoder, 512 x 128-token fp16 batch and the ragged 64..200 batch -> ms per batch (median, min); outputs must be identical
scan: search steps (scan + merge, the twin that trusts pmax, as bench.py issues them) on the first 1M / 4M / 32M rows of one slab -> ms per step
The builds come from `scripts/build_variant.sh <git rev | worktree dir> <out.so>` (product flags)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atlas_amd import _lib, retrievers

what = sys.argv[1]
libs, rounds = [], 5
for a in sys.argv[2:]:
    if "=" in a:
        name, path = a.split("=", 1)
        libs.append((name, _lib._bind(os.path.abspath(path))))
    else:
        rounds = int(a)

if what in ("enc", "both"):
    m = retrievers.Contriever(retrievers.BertConfigLite()).half().eval().cuda().requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    NB = 512

    def batch(lens, L_):
        ids = torch.randint(1000, 30522, (NB, L_), generator=g)
        mask = (torch.arange(L_)[None, :] < lens[:, None]).long()
        return (ids * mask).cuda(), mask.cuda()

    work = {"full 512x128": batch(torch.full((NB,), 128), 128)}
    lens = torch.randint(64, 201, (NB,), generator=g)
    work["ragged 64..200"] = batch(lens, int(lens.max()))
    out = torch.empty((NB, 768), dtype=torch.float16, device="cuda")
    for wname, (ids, mask) in work.items():
        res = {n: [] for n, _ in libs}
        ref = None
        for r in range(rounds):
            for n, h in libs:
                m._library = h
                m.embed_into(out, ids, mask)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(4):
                    m.embed_into(out, ids, mask)
                torch.cuda.synchronize()
                res[n].append((time.perf_counter() - t) / 4 * 1e3)
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(out, ref):
                    print(f"!! build {n} differs from {libs[0][0]} on {wname}: max |d| = {(out.float() - ref.float()).abs().max().item():.3e}", flush=True)
        for n, t in res.items():
            print(f"{wname:16s} {n:24s}: {np.median(t):7.3f} ms (min {min(t):7.3f})  {NB / np.median(t) * 1e3:8.0f} passages/s", flush=True)
    del m, out, work
    torch.cuda.empty_cache()

if what in ("scan", "both"):
    B, k, D = 64, 40, 768
    NMAX = int(os.environ.get("LIB_AB_ROWS", 32_000_000))
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((NMAX, D), dtype=torch.float16, device="cuda")
    for r0 in range(0, NMAX, 1_000_000):
        n = min(1_000_000, NMAX - r0)
        x = torch.randn((n, D), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    del x
    q32 = torch.randn((B, D), generator=torch.Generator(device="cuda").manual_seed(99), device="cuda")
    out_s = torch.empty((B, k), dtype=torch.float16, device="cuda")
    out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    out_st = torch.empty(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    pm = torch.zeros(1, dtype=torch.float32, device="cuda")
    for N in [n for n in (1_000_000, 4_000_000, 32_000_000) if n <= NMAX]:
        reps = 200 if N <= 4_000_000 else 30
        assert libs[0][1].atlas_slab_pmax(slab.data_ptr(), N, D, pm.data_ptr(), stream) == 0
        pmax = float(pm.item())
        for qn, q, qdt in (("f32 queries", q32, _lib.DT_F32), ("f16 queries", q32.half(), _lib.DT_F16)):
            res = {n: [] for n, _ in libs}
            ref = None
            wss = {n: torch.zeros(h.atlas_scan_topk_workspace_bytes(N, B, D, k), dtype=torch.uint8, device="cuda") for n, h in libs}   # (the head of a workspace holds state: zero before its first use)
            for r in range(rounds):
                for n, h in libs:
                    ws = wss[n]

                    def call():
                        rc = h.atlas_scan_topk_flags(q.data_ptr(), qdt, slab.data_ptr(), N, B, D, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), stream, None, None, _lib.SCAN_TRUST_PMAX)
                        assert rc == 0, rc
                    for _ in range(5):
                        call()
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for _ in range(reps):
                        call()
                    torch.cuda.synchronize()
                    res[n].append((time.perf_counter() - t) / reps * 1e3)
                    assert int(out_st[_lib.ST_FLAGS]) == 0
                    cur = (out_s.clone(), out_i.clone())
                    if ref is None:
                        ref = cur
                    elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
                        print(f"!! build {n} differs from {libs[0][0]} at {N} rows", flush=True)
            for n, t in res.items():
                med = float(np.median(t))
                print(f"scan {N:>9d} rows, {qn}  {n:24s}: {med:7.4f} ms per step (min {min(t):7.4f})  step / 8 TB/s = {N * 1536 / (med * 1e-3) / 8e12:.4f}", flush=True)
