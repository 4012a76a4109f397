#!/bin/bash
TAG=${1:-q01}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_HIT\[[^ ]*\|TCC_[A-Z_0-9]*_sum\|SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_WAIT[A-Z_]*\|SQ_VALU_MFMA[A-Z_]*\|SQ_ACTIVE_INST_[A-Z]*\|MfmaUtil\|GRBM_GUI_ACTIVE" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/$OUT/avail.txt
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/enc_pmc_run.py 2 > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1; echo "pass $i ($pmc) rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_bt" not in k and "attention" not in k: continue
        short = k[k.index("<"):k.index(">")+1] if "<" in k else k[:40]
        agg[(short[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-42s %-30s n=%3d mean=%.4g" % (k, c, len(v), sum(v)/len(v)))
PY
