"""Soak of the DMA-staged 64-query pass (dscan_kernel.h) against the register-fed one (scan_kernel.h), tuning build, one process: many searches with
fresh random queries per shard -- sizes that take the tile pool, k up to 256 (wide candidate bands: flushes and compactions mid-scan), duplicated and
near-duplicated rows (ties), both twins, the three query dtypes -- every result compared bit for bit with the other kernel's (a schedule race, a lost
pool tile or a dropped candidate shows up as an intermittent mismatch) and, every 25th search, with the MFMA-free exact path.

    python tools/scan_soak.py [searches per shard, default 300]
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import ctypes, sys, time
import numpy as np
import torch
from atlas_amd import HipDistributedIndex, _lib
from scan_policy_common import shard

L.atlas_tune_set_scan_dma.argtypes, L.atlas_tune_set_scan_dma.restype = [ctypes.c_int], None
per = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(5)
g = torch.Generator(device="cuda").manual_seed(6)
stream = torch.cuda.current_stream().cuda_stream
dts = [(torch.float32, _lib.DT_F32), (torch.float16, _lib.DT_F16), (torch.bfloat16, _lib.DT_BF16)]
total = bad = fallbacks = 0
t_start = time.time()
for N, hard in ((700_001, False), (1_000_000, True), (2_621_440, False), (8_000_000, False), (8_000_000, True)):
    slab = shard(N, seed=N % 97)
    if hard:       # blocks of duplicated rows and of rows that differ in one element: many exact and near ties around every threshold
        n = N // 8
        slab[N // 2: N // 2 + n] = slab[:n]
        slab[N // 4: N // 4 + 4096, 5] += 0.0005
    pmax = float(slab[: min(N, 2_000_000)].float().norm(dim=1).max()) * 1.01
    ref = HipDistributedIndex(); ref._set_slab(slab)
    ws = torch.zeros(int(L.atlas_scan_topk_workspace_bytes(N, 64, 768, 256)), dtype=torch.uint8, device="cuda")
    for it in range(per):
        B = int(rng.choice([64, 64, 64, 33, 1]))
        k = int(rng.choice([40, 40, 256, 100, 5]))
        tdt, code = dts[it % 3]
        q = (torch.randn((B, 768), generator=g, device="cuda") * float(rng.choice([1.0, 0.2, 5.0]))).to(tdt)
        flags = _lib.SCAN_TRUST_PMAX if it % 2 == 0 else 0
        res = {}
        for mode in (1, 0):
            L.atlas_tune_set_scan_dma(mode)
            out_s = torch.zeros((B, k), dtype=torch.float16, device="cuda"); out_i = torch.zeros((B, k), dtype=torch.int64, device="cuda")
            out_st = torch.zeros(_lib.STATUS_HEADER + B, dtype=torch.int32, device="cuda")
            rc = L.atlas_scan_topk_flags(q.data_ptr(), code, slab.data_ptr(), N, B, 768, k, pmax, out_s.data_ptr(), out_i.data_ptr(), out_st.data_ptr(),
                                         ws.data_ptr(), ws.numel(), stream, None, None, flags)
            assert rc == 0, rc
            torch.cuda.synchronize()
            res[mode] = (out_s, out_i, out_st[: _lib.STATUS_HEADER].tolist())
        total += 1
        flags_ok = res[1][2][_lib.ST_FLAGS] in (0, _lib.F_FALLBACK) and res[0][2][_lib.ST_FLAGS] in (0, _lib.F_FALLBACK)
        # (a search whose candidates outgrow the lists -- one query with k = 256 on 8M rows brings ~125k of the 131 072 the merge takes -- is handed to the exact
        #  path, ATLAS_F_FALLBACK, by whichever kernel's thresholds tightened a little later in that call: the outputs of a flagged call are not results)
        handed = res[1][2][_lib.ST_FLAGS] != 0 or res[0][2][_lib.ST_FLAGS] != 0
        fallbacks += int(handed)
        same = handed or (torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]))
        exact_ok = True
        if it % 25 == 0 and res[1][2][_lib.ST_FLAGS] == 0:      # (a query the scan hands to the exact path -- ATLAS_F_FALLBACK -- has no result in this call's output)
            es, ei = ref._exact_topk(q, k)
            exact_ok = torch.equal(res[1][0], es) and torch.equal(res[1][1], ei)
        if not (same and flags_ok and exact_ok):
            bad += 1
            print(f"MISMATCH N={N} hard={hard} it={it} B={B} k={k} dtype={tdt} flags={flags}: same={same} status dma={res[1][2]} reg={res[0][2]} exact_ok={exact_ok}", flush=True)
    print(f"N={N:9d} hard={hard!s:5s}: {per} searches, mismatches so far {bad} of {total}  (candidates of the last search: dma {res[1][2][_lib.ST_N_CANDIDATES]}, register-fed {res[0][2][_lib.ST_N_CANDIDATES]}; "
          f"fallback queries dma {res[1][2][_lib.ST_N_FALLBACK]})  [{time.time() - t_start:.0f} s]", flush=True)
    del slab, ref, ws
    torch.cuda.empty_cache()
L.atlas_tune_set_scan_dma(1)
print(f"TOTAL {total} searches x 2 kernels, {bad} mismatches, {fallbacks} searches handed to the exact path by at least one kernel")
sys.exit(1 if bad else 0)
