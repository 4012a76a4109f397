#!/bin/bash
# round 2 profiles: rocprofv3 --kernel-trace --stats of the default bench command; PMC FETCH_SIZE passes (own runs) at 1M / 4M / 32M rows
OUT=gpurun_out/r02p; mkdir -p $OUT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w tools/microbench.hip -o tools/libatlas_mb.so 2>&1 | tail -2
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1); echo "rocprof rc=$?"
f=$(find $OUT/prof_default -name "*kernel_stats*.csv" | head -1); cp $f $OUT/bench_default_kernel_stats.csv; head -30 $f | cut -c1-160
for n in 1000000 4000000 32000000; do
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $n > $GRAFT_REPO_ROOT/$OUT/pmc_$n.log 2>&1); echo "pmc $n rc=$?"
  cp $(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1) $OUT/pmc_${n}_fetch_counter_collection.csv
done
python tools/pmc_summarize.py $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
rm -rf $OUT/prof_default $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000
