"""summarise the kernel trace of tools/gemm_diag.py: median duration of the four GEMMs per diag mode (passes 2-4 of each mode)"""
import csv, sys, statistics as st
path, modes = sys.argv[1], sys.argv[2:] or ["0", "1", "2"]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
g = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if any(k in r["Kernel_Name"] for k in ("gemm_pp", "gemm_co", "gemm_wr"))]
per_mode = len(g) // len(modes)
other = {}
for r in rows:
    n = r["Kernel_Name"]
    if "gemm" not in n:
        other.setdefault(n.split("(")[0][:48], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for i, md in enumerate(modes):
    seq = g[i * per_mode + per_mode // 4:(i + 1) * per_mode]       # drop the first pass of the mode
    qkv, op, f1, f2 = seq[0::4], seq[1::4], seq[2::4], seq[3::4]
    med = [st.median([d for _, d in x]) for x in (qkv, op, f1, f2)]
    print("diag %s: qkv %.1f  out-proj %.1f  ffn1 %.1f  ffn2 %.1f   sum %.1f us" % (md, *med, sum(med)))
for k, v in sorted(other.items()):
    print("   %-50s n=%4d median %.1f us" % (k, len(v), st.median(v)))
