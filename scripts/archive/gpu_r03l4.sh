#!/bin/bash
OUT=gpurun_out/r03l; mkdir -p $OUT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w tools/microbench.hip -o tools/libatlas_mb.so 2>&1 | tail -2
for v in head park; do
  for c in "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $c | cut -d' ' -f1)
    (cd /tmp && ATLAS_HIP_SO=$GRAFT_REPO_ROOT/tools/ab/$v.so timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_${v}_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py 4000000 > $GRAFT_REPO_ROOT/$OUT/pmc_${v}_$tag.log 2>&1)
    f=$(find $OUT/pmc_${v}_$tag -name "*counter_collection.csv" | head -1)
    python - "$f" "$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "scan_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in acc.items():
    vals = [sum(v) for v in d.values()]
    print(sys.argv[2], c, "per launch (last):", vals[-1], "launches", len(vals))
PY
    rm -rf $OUT/pmc_${v}_$tag
  done
done 2>&1 | tee $OUT/pmc_compare.txt
