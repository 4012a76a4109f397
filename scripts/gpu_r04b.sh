#!/bin/bash
# round 4, session b: where the GEMM-shaped scan's time goes: per-kernel durations (rocprofv3 kernel trace) at 4M rows x 256 / 512 queries and on a
# slab that fits the Infinity Cache (131 072 rows), then MFMA-busy / LDS-conflict / FETCH_SIZE counters (own passes)
OUT=gpurun_out/r04b
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "4000000 256" "4000000 512" "131072 256" "131072 512"; do
  set -- $cfg
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_$1_$2 -o t -- python $R/tools/gscan_run.py $1 $2 10 > $R/$OUT/kt_$1_$2.log 2>&1)
  echo "== $cfg" | tee -a $OUT/summary.log
  grep -E "gscan|gtheta|gprep|merge_rescore" $OUT/kt_$1_$2/t_kernel_stats.csv | cut -c1-200 | tee -a $OUT/summary.log
done
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $R/$OUT/pmc_mfma -o t -- python $R/tools/gscan_run.py 4000000 512 4 > $R/$OUT/pmc_mfma.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$OUT/pmc_lds -o t -- python $R/tools/gscan_run.py 4000000 512 4 > $R/$OUT/pmc_lds.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -o t -- python $R/tools/gscan_run.py 4000000 512 4 > $R/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch256 -o t -- python $R/tools/gscan_run.py 4000000 256 4 > $R/$OUT/pmc_fetch256.log 2>&1)
python - <<'PY' | tee -a $OUT/summary.log
import csv, glob, collections
for d in ("pmc_mfma", "pmc_lds", "pmc_fetch", "pmc_fetch256"):
    for f in glob.glob(f"gpurun_out/r04b/{d}/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "gscan_kernel" not in kn: continue
            acc[kn[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kn, cs in acc.items():
            print(d, kn, {c: sum(v) / len(v) for c, v in cs.items()}, "launches", {c: len(v) for c, v in cs.items()})
PY
