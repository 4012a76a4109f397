#!/bin/bash
TAG=${1:-p01}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
N=${2:-4000000}
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $N > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1); echo "pmc FETCH rc=$?"
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $N > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1); echo "pmc WRITE rc=$?"
ls $OUT/pmc_fetch; tail -3 $OUT/pmc_fetch.log
python - <<PY
import csv, glob, collections
for tag in ("fetch", "write"):
    fs = glob.glob("$OUT/pmc_%s/*counter_collection.csv" % tag)
    if not fs: print("no counter csv for", tag); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(tag, "%-62s %-12s n=%3d mean=%.1f" % (k, c, len(v), sum(v)/len(v)))
PY
