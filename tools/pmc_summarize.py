"""Turn the rocprofv3 --pmc FETCH_SIZE passes of tools/pmc_run.py into profiles/pmc_traffic.json (what bench.py reports as
roofline.traffic). FETCH_SIZE on gfx950 under-reports wide reads ~2x (MI355X_MICROARCH.md §HBM), so the scan's counter is calibrated
against stream_kernel<1,..> in the SAME pass, which reads exactly rows x 1536 bytes with the scan's access pattern:
    traffic = FETCH_SIZE(scan) x known_bytes / FETCH_SIZE(stream).
The file records the sha256 of the CODE of csrc/scan_kernel.h + csrc/merge_kernel.h + csrc/atlas_hip.hip (the kernel AND its launch plan: pool split, grid;
comments and blank lines stripped: atlas_amd._lib.scan_sources_sha256): bench.py only quotes a traffic figure measured on the code it runs.

    python tools/pmc_summarize.py gpurun_out/r02p/pmc_1000000 gpurun_out/r02p/pmc_4000000 gpurun_out/r02p/pmc_32000000
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atlas_amd import _lib  # noqa: E402
out = {"note": "HBM read traffic of ONE scan_kernel launch from rocprofv3 --pmc FETCH_SIZE (own pass, --kernel-trace only), calibrated in the same pass "
               "against stream_kernel<1,4,8> (exactly rows*1536 B, the scan's access pattern): traffic = FETCH_SIZE_scan * known_bytes / FETCH_SIZE_stream. "
               "Raw CSVs: profiles/r04/pmc_*_fetch_counter_collection.csv.",
       "kernel": "scan_kernel<16,1,8>",
       "sources_sha256": _lib.scan_sources_sha256(),
       "per_rows": {}}
for d in sys.argv[1:]:
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == "FETCH_SIZE"]
    scan = [float(r["Counter_Value"]) for r in rows if "scan_kernel" in r["Kernel_Name"]]
    cal = [float(r["Counter_Value"]) for r in rows if "stream_kernel<1" in r["Kernel_Name"]]
    grid = [int(r["Grid_Size"]) for r in rows if "scan_kernel" in r["Kernel_Name"]]
    n = int(os.path.basename(d.rstrip("/")).split("_")[-1])
    scan_kib, cal_kib = sum(scan[1:]) / len(scan[1:]), sum(cal) / len(cal)           # first scan launch: warm-up
    out["per_rows"][str(n)] = {"fetch_size_scan_kib": scan_kib, "fetch_size_calib_kib": cal_kib, "calib_bytes": n * 1536,
                               "traffic_bytes": int(scan_kib * (n * 1536) / cal_kib), "scan_launches": len(scan), "ratio_to_algorithmic": scan_kib / cal_kib}
    print(n, out["per_rows"][str(n)])
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
