#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench workload (32M rows only: the shard sweep would mix other sizes into the scan's average)
OUT=gpurun_out/r02p; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --shard-sweep '' --refresh-stream-seconds 0 > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1); echo "rocprof rc=$?"
f=$(find $OUT/prof_default -name "*kernel_stats*.csv" | head -1); cp $f $OUT/bench_default_kernel_stats.csv; grep '"metric"' $OUT/prof_default.log > $OUT/bench_under_rocprof.json
rm -rf $OUT/prof_default
