"""LDS bank-conflict share per encoder / scan kernel from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass (tools/enc_pmc_run.py):   python tools/pmc_lds_summarize.py <dir>"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("gemm_", "attention_kernel", "ln_kernel", "scan_kernel", "stream_dma_kernel")): continue
        short = (k[:k.index("(")] if "(" in k else k).replace("void ", "")
        agg[short[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    m = {n: sum(v[1:] if len(v) > 2 else v) / max(1, len(v[1:] if len(v) > 2 else v)) for n, v in c.items()}
    print("%-44s" % k, "  ".join("%s %.4g" % (n, m[n]) for n in sorted(m)), "  conflict share of LDS cycles %.2f %%" % (100.0 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1))))
