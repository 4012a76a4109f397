#!/bin/bash
# one PMC pass (FETCH_SIZE, own run, --kernel-trace only) over the calibration stream + the product search at 32M rows
export TMPDIR=/tmp; OUT=gpurun_out/${1:-pmc2}; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py ${2:-32000000} > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1); echo "rc=$?"
python - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/pmc_fetch/*counter_collection.csv")
rows = list(csv.DictReader(open(fs[0])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print("%-72s %-12s n=%3d mean=%.1f" % (k, c, len(v), sum(v)/len(v)))
PY
