"""Per-layer GEMM times of the 512 x 128-token fp16 refresh batch from a rocprofv3 kernel trace of `tools/gemm_diag.py <cfg>:0 ...`
(median over layers and passes, first pass of every configuration dropped).   python tools/gemm_layer_report.py trace.csv 9:0 4:0"""
import csv, sys, statistics as st
path, modes = sys.argv[1], sys.argv[2:]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
# split the trace into the modes: every mode runs 4 passes of 12 layers; a pass starts with embed_ln_kernel
starts = [i for i, (n, _) in enumerate(ev) if "embed_ln_kernel" in n]
assert len(starts) == 4 * len(modes), (len(starts), modes)
for mi, md in enumerate(modes):
    seg = ev[starts[4 * mi + 1]: starts[4 * mi + 4] if 4 * mi + 4 < len(starts) else len(ev)]     # passes 2-4
    g = [(n, d) for n, d in seg if "gemm_" in n]
    # the persistent kernel of rounds 3-4 (and tuning cfg 10) launched the QKV projection twice (q|k, then V with the V^T epilogue: an EPI 4 kernel
    # in the trace); round 5's cfg 9 launches it once
    per_layer = 5 if any("gemm_pt" in n and ", 4>" in n for n, _ in g) else 4
    names = ["q|k", "V", "out-proj", "FFN-1", "FFN-2"] if per_layer == 5 else ["QKV", "out-proj", "FFN-1", "FFN-2"]
    med = [st.median([d for _, d in g[j::per_layer]]) for j in range(per_layer)]
    oth = {}
    for n, d in seg:
        if "gemm_" not in n:
            oth.setdefault(n.split("(")[0].replace("void ", "")[:32], []).append(d)
    print(f"cfg:diag {md}: " + "  ".join(f"{a} {b:.1f}" for a, b in zip(names, med)) + f"   GEMMs per layer {sum(med):.1f} us;  " +
          "  ".join(f"{k} {st.median(v):.1f}" for k, v in sorted(oth.items()) if "ln_kernel" in k or "attention" in k))
