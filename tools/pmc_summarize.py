"""Turn the rocprofv3 --pmc FETCH_SIZE passes of tools/pmc_run.py into profiles/pmc_traffic.json (what bench.py reports as
roofline.traffic). FETCH_SIZE on gfx950 under-reports wide reads ~2x (MI355X_MICROARCH.md §HBM), so the scan's counter is calibrated
against a stream in the SAME pass that reads exactly rows x 1536 bytes with the scan's access pattern -- stream_dma_kernel (LDS-DMA nt, 8 rows x
128 B per wave instruction) for dscan_kernel.h, the 64-query pass since round 6; stream_kernel<1,..> (register loads in the MFMA fragment shape)
for scan_kernel.h --:
    traffic = FETCH_SIZE(scan) x known_bytes / FETCH_SIZE(stream).
The file records the sha256 of the CODE of csrc/dscan_kernel.h + csrc/scan_kernel.h + csrc/merge_kernel.h + csrc/atlas_hip.hip (the kernel AND its launch plan: pool split, grid;
comments and blank lines stripped: atlas_amd._lib.scan_sources_sha256): bench.py only quotes a traffic figure measured on the code it runs.

    python tools/pmc_summarize.py gpurun_out/r02p/pmc_1000000 gpurun_out/r02p/pmc_4000000 gpurun_out/r02p/pmc_32000000
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atlas_amd import _lib  # noqa: E402
out = {"note": "HBM read traffic of ONE 64-query scan launch from rocprofv3 --pmc FETCH_SIZE (own pass, --kernel-trace only), calibrated in the same pass "
               "against a stream of exactly rows*1536 B in the scan's own access pattern (dscan_kernel: stream_dma_kernel, LDS-DMA nt, 8 rows x 128 B per wave "
               "instruction): traffic = FETCH_SIZE_scan * known_bytes / FETCH_SIZE_stream. Raw CSVs: profiles/r06/pmc_*_fetch_counter_collection.csv.",
       "kernel": None,
       "sources_sha256": _lib.scan_sources_sha256(),
       "per_rows": {}}
for d in sys.argv[1:]:
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == "FETCH_SIZE"]
    dma = any("dscan_kernel" in r["Kernel_Name"] for r in rows)
    kname, cname = ("dscan_kernel", "stream_dma_kernel") if dma else ("scan_kernel", "stream_kernel<1")
    out["kernel"] = "dscan_kernel<nt> (trusted pmax)" if dma else "scan_kernel<16,1,8>"
    scan = [float(r["Counter_Value"]) for r in rows if kname in r["Kernel_Name"]]
    cal = [float(r["Counter_Value"]) for r in rows if cname in r["Kernel_Name"]]
    grid = [int(r["Grid_Size"]) for r in rows if kname in r["Kernel_Name"]]
    n = int(os.path.basename(d.rstrip("/")).split("_")[-1])
    scan_kib, cal_kib = sum(scan[1:]) / len(scan[1:]), sum(cal) / len(cal)           # first scan launch: warm-up
    out["per_rows"][str(n)] = {"fetch_size_scan_kib": scan_kib, "fetch_size_calib_kib": cal_kib, "calib_bytes": n * 1536,
                               "traffic_bytes": int(scan_kib * (n * 1536) / cal_kib), "scan_launches": len(scan), "ratio_to_algorithmic": scan_kib / cal_kib, "calibration_kernel": cname}
    print(n, out["per_rows"][str(n)])
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
