#!/bin/bash
# query-batch profile: per-GEMM durations of one forward for several small-batch GEMM configurations
export TMPDIR=/tmp
for cfg in 3 5; do
  (cd /tmp && ATLAS_GEMM_CFG=$cfg rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/qp$cfg -o t -- python $GRAFT_REPO_ROOT/tools/enc_query_prof.py ${1:-fp16} > /dev/null 2>&1)
  python - <<PY
import csv, statistics as st
rows=list(csv.DictReader(open("gpurun_out/qp$cfg/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=[r for r in rows if "at::" not in r["Kernel_Name"]]
idx=[i for i,r in enumerate(rows) if "count_kernel" in r["Kernel_Name"]]
seg=rows[idx[-1]:]
d=lambda r:(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
g=[(r["Kernel_Name"],d(r)) for r in seg if "gemm" in r["Kernel_Name"]]
qkv=[t for i,(n,t) in enumerate(g) if i%4==0]; op=[t for i,(n,t) in enumerate(g) if i%4==1]; f1=[t for i,(n,t) in enumerate(g) if i%4==2]; f2=[t for i,(n,t) in enumerate(g) if i%4==3]
tot=sum(d(r) for r in seg)
print("cfg $cfg: forward %.0f us  qkv %.1f outproj %.1f ff1 %.1f ff2 %.1f  (gemm sum/layer %.1f)"%(tot, st.mean(qkv),st.mean(op),st.mean(f1),st.mean(f2),st.mean(qkv)+st.mean(op)+st.mean(f1)+st.mean(f2)))
PY
done
