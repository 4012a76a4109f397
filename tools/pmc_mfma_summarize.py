"""MFMA-pipe utilisation per kernel from rocprofv3 --pmc passes (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE in one pass):
util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)   (busy cycles are summed over all SIMDs, GUI_ACTIVE over the
8 XCDs; a 16x16x32 f16 MFMA keeps its SIMD's pipe busy 16 cycles, MI355X_MICROARCH.md PMC notes).

    python tools/pmc_mfma_summarize.py <dir with *counter_collection.csv> [...]
"""
import collections, csv, glob, sys

for d in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(t in k for t in ("scan_kernel", "gemm_", "attention_kernel", "merge_rescore", "ln_kernel")):
                continue
            short = k[:k.index("(")] if "(" in k else k
            short = short.replace("void ", "").replace("atlas::", "")
            agg[short[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"# {d}")
    for k, c in sorted(agg.items()):
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
        if not busy or not act:
            continue
        # drop the cold first launch of a kernel when there are several
        b = busy[1:] if len(busy) > 2 else busy
        a = act[1:] if len(act) > 2 else act
        mb, ma = sum(b) / len(b), sum(a) / len(a)
        extra = ""
        if "SQ_BUSY_CYCLES" in c:
            extra = "  SQ_BUSY_CYCLES %.4g" % (sum(c["SQ_BUSY_CYCLES"]) / len(c["SQ_BUSY_CYCLES"]))
        print("%-46s launches %3d  MFMA_BUSY %.4g  GUI_ACTIVE %.4g (%.0f cycles per XCD)  MFMA pipe utilisation %.1f %%%s"
              % (k, len(busy), mb, ma, ma / 8, 100.0 * mb / (ma / 8 * 1024), extra))
