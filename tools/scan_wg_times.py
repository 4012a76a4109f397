"""Per-workgroup WALL-CLOCK stamps of the scan kernel (tuning build): when every workgroup starts, has its query image, finishes its
first tile, its last tile and its hand-over, on the chip-wide 100 MHz clock (10 ns ticks) -- start skew, per-CU streaming rate,
end skew, per XCD. Shows where the size-independent part of the scan time goes.

    python tools/scan_wg_times.py 1000000 4000000
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402  (tuning build of the library, hooks bound)
import sys
import numpy as np
import torch
from atlas_amd import HipDistributedIndex

for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000]:
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 250_000):
        n = min(250_000, N - r0); x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0+n] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((64, 768), device="cuda")
    idx = HipDistributedIndex(); idx._set_slab(slab)
    for _ in range(5): idx._compute_scores_and_indices(q, 40)
    dbg = torch.zeros(2048 + 256 * 120, dtype=torch.int64, device="cuda")
    res = []
    for rep in range(6):
        dbg.zero_()
        L.atlas_tune_set_scan_stamps(dbg.data_ptr())
        idx._compute_scores_and_indices(q, 40); torch.cuda.synchronize()
        L.atlas_tune_set_scan_stamps(None)
        raw = dbg.cpu().numpy().astype(np.float64) * 0.01                    # microseconds
        t = raw[:2048].reshape(256, 8)
        t0 = t[:, 0].min()
        tiles = raw[2048:].reshape(256, 120)
        res.append(t - t0)
        last_tiles = np.where(tiles > 0, tiles - t0, np.nan)
    f = lambda a: "min %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    for rep, t in enumerate(res[-3:]):
        rows_wg = -(-(N // 16) // 256) * 16
        print(f"N={N} rep {rep}: kernel span (first entry -> last hand-over) = {t[:, 5].max():.2f} us; ideal at 6.05 TB/s = {N * 1536 / 6.05e6:.2f} us")
        print("  entry (abs)            ", f(t[:, 0]))
        print("  image barrier passed   ", f(t[:, 1] - t[:, 0]), "(since entry)")
        print("  threshold published    ", f(t[:64, 6] - t[:64, 0]), "(since entry, workgroups 0..63)")
        print("  thresholds collected   ", f(t[:, 7] - t[:, 0]), "(since entry)")
        print("  query image in LDS     ", f(t[:, 2] - t[:, 0]), "(since entry)")
        print("  first tile done        ", f(t[:, 3] - t[:, 0]), "(since entry)")
        print("  loop (image -> last)   ", f(t[:, 4] - t[:, 2]), "(duration)")
        print("  per-WG stream rate GB/s", f(rows_wg * 1536 / (t[:, 4] - t[:, 2]) / 1e3))
        print("  last tile done (abs)   ", f(t[:, 4]))
        print("  hand-over              ", f(t[:, 5] - t[:, 4]), "(duration)")
        print("  end (abs)              ", f(t[:, 5]))
        print("  mean loop per XCD (wg % 8):", [round(float((t[x::8, 4] - t[x::8, 2]).mean()), 1) for x in range(8)])
        print("  mean entry per XCD        :", [round(float(t[x::8, 0].mean()), 2) for x in range(8)])
        order = np.argsort(t[:, 5])
        print("  last 8 WGs to finish:", order[-8:].tolist(), "loop us:", np.round(t[order[-8:], 4] - t[order[-8:], 2], 1).tolist(),
              "hand-over us:", np.round(t[order[-8:], 5] - t[order[-8:], 4], 1).tolist())
        # how much of the chip is still streaming over time: number of WGs still in their loop at the p50 / p90 / max end
        ends = np.sort(t[:, 4])
        print("  WGs still looping at t = p50 end: %d, p90: %d; time from p50 end to last end: %.2f us" % (
            (t[:, 4] > ends[127]).sum(), (t[:, 4] > ends[230]).sum(), ends[-1] - ends[127]))
    # per-tile picture of the last repetition: tile period (us) of the median workgroup vs the 8 slowest, tile by tile
    t = res[-1]
    order = np.argsort(t[:, 4])
    nt = int(np.isfinite(last_tiles).sum(axis=1).max())       # (workgroups draw different numbers of pool tiles)
    print('  tiles per workgroup: min %d max %d' % (np.isfinite(last_tiles).sum(axis=1).min(), nt))
    per = np.diff(np.concatenate([t[:, 2:3], last_tiles[:, :nt]], axis=1), axis=1)      # [wg][tile] period
    med = np.nanmedian(per, axis=0)
    print(f"N={N}: tile periods (us), {nt} tiles; median over workgroups:", np.round(med, 1).tolist())
    for w in order[-6:].tolist() + order[:2].tolist():
        print(f"   wg {w:3d} (xcd {w % 8}) loop {t[w, 4] - t[w, 2]:7.1f} us:", np.round(per[w], 1).tolist())
    slow = order[-32:]
    print("   mean period of the 32 slowest WGs minus the median WG, per tile:", np.round(np.nanmean(per[slow], axis=0) - med, 2).tolist())
    del slab, idx
    torch.cuda.empty_cache()
