#!/bin/bash
# bulk-GEMM session: parity under a forced configuration, then the refresh leg per configuration (per-GEMM medians)
OUT=gpurun_out/${1:-g01}; mkdir -p $OUT; export TMPDIR=/tmp
CFGS=${2:-"8 4"}
for cfg in $CFGS; do
  ATLAS_GEMM_CFG=$cfg timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q --no-header -x -p no:cacheprovider -k "not model_precision" 2>&1 | tail -1
  (cd /tmp && ATLAS_GEMM_CFG=$cfg rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/c$cfg -o t -- python $GRAFT_REPO_ROOT/bench.py --passages 1000000 --steps 3 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/c$cfg.log 2>&1)
  python - <<PY
import csv, json, statistics as st
lines=[l for l in open("$OUT/c$cfg.log").read().splitlines() if l.startswith("{")]
d=json.loads(lines[-1])["refresh"]
rows=list(csv.DictReader(open("$OUT/c$cfg/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"][r["Kernel_Name"].index("<"):r["Kernel_Name"].index(">")+1],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if "gemm" in r["Kernel_Name"]]
seq=seq[:4*12*7]          # the fixed-length leg (7 batches) comes first
e2=[d for k,d in seq if k.startswith("<F16, 2")]
op=e2[0::2]; ff2=e2[1::2]
e1=[d for k,d in seq if k.startswith("<F16, 1")]; e3=[d for k,d in seq if k.startswith("<F16, 3")]
print("cfg $cfg refresh", round(d["value"]), "passages/s", round(d["roofline"]["achieved"],1), "TF   qkv %.1f outproj %.1f ff1 %.1f ff2 %.1f  sum %.1f"%(st.median(e3),st.median(op),st.median(e1),st.median(ff2),st.median(e3)+st.median(op)+st.median(e1)+st.median(ff2)), " ragged", round(d["ragged"]["value"]))
PY
done
