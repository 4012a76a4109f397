"""Where the time between the slowest scan workgroup's hand-over and the search's last output goes, with the merge as a launch of its own
(fused 0) and inside the scan's last workgroups (fused 1): wall-clock stamps (100 MHz, chip-wide) of the tuning build.
    python tools/fused_timeline.py 1000000 4000000
"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _tune import L  # noqa: E402
import sys
import numpy as np
import torch
from atlas_amd import HipDistributedIndex

for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000]:
    g = torch.Generator(device="cuda").manual_seed(1)
    slab = torch.empty((N, 768), dtype=torch.float16, device="cuda")
    for r0 in range(0, N, 250_000):
        n = min(250_000, N - r0); x = torch.randn((n, 768), generator=g, device="cuda")
        slab[r0:r0 + n] = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((64, 768), device="cuda")
    for fused in (0, 1, 0, 1):
        L.atlas_tune_set_scan_fused(fused)
        idx = HipDistributedIndex(); idx._set_slab(slab)
        for _ in range(5): idx._compute_scores_and_indices(q, 40)
        sd = torch.zeros(2048 + 256 * 120, dtype=torch.int64, device="cuda")
        md = torch.zeros(8, dtype=torch.int64, device="cuda")
        rows = []
        for rep in range(5):
            sd.zero_(); md.zero_()
            L.atlas_tune_set_scan_stamps(sd.data_ptr()); L.atlas_tune_set_merge_stamps(md.data_ptr())
            idx._compute_scores_and_indices(q, 40); torch.cuda.synchronize()
            L.atlas_tune_set_scan_stamps(None); L.atlas_tune_set_merge_stamps(None)
            t = sd.cpu().numpy()[:2048].reshape(256, 8).astype(np.float64) * 0.01
            m = md.cpu().numpy().astype(np.float64) * 0.01
            t0 = t[:, 0].min()
            last5 = t[:, 5].max() - t0
            rows.append([t[:, 4].max() - t0, last5] + [m[i] - t0 - last5 for i in range(7)])
        r = np.median(np.array(rows), axis=0)
        names = ["merge q0 entered", "table + heads", "keys in LDS", "threshold", "band", "rescored", "ranked + written"]
        print(f"N={N} fused={fused}: last tile done {r[0]:8.2f} us, last hand-over done {r[1]:8.2f} us after the first entry; then, in us after that hand-over: "
              + "  ".join(f"{n} {v:6.2f}" for n, v in zip(names, r[2:])), flush=True)
L.atlas_tune_set_scan_fused(0)
