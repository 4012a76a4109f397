#!/bin/bash
# encoder session: parity tests, per-kernel times of the refresh leg, batch-shape timings
OUT=gpurun_out/${1:-e03}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_encoder_golden.py tests/test_gpu_end_to_end.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | tail -3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --passages 1000000 --steps 3 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/bench.log 2>&1)
python - <<PY
import csv, json, statistics as st
lines=[l for l in open("$OUT/bench.log").read().splitlines() if l.startswith("{")]
d=json.loads(lines[-1])["refresh"]
print("refresh", round(d["value"]), "passages/s", round(d["roofline"]["achieved"],1), "TF")
rows=list(csv.DictReader(open("$OUT/prof/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"][r["Kernel_Name"].index("<"):r["Kernel_Name"].index(">")+1],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if "gemm" in r["Kernel_Name"]]
e2=[d for k,d in seq if k.startswith("<F16, 2")]
op=e2[0::2]; ff2=e2[1::2]
e1=[d for k,d in seq if k.startswith("<F16, 1")]; e3=[d for k,d in seq if k.startswith("<F16, 3")]
print("   qkv %.1f outproj %.1f ff1 %.1f ff2 %.1f  sum %.1f us"%(st.median(e3),st.median(op),st.median(e1),st.median(ff2),st.median(e3)+st.median(op)+st.median(e1)+st.median(ff2)))
for r in csv.DictReader(open("$OUT/prof/t_kernel_stats.csv")):
    n=r["Name"]
    if any(k in n for k in ("attention","ln_kernel","pool","embed_ln","count_k","pack_k")):
        print("   %-60s calls=%4s avg=%9.1f us"%(n[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
for a in "fp16 trim" "fp32 trim" "bf16 trim"; do timeout 300 python tools/enc_time.py $a 2>&1 | grep -v amdgpu.ids | tee -a $OUT/enc_time.log; done
