#!/bin/bash
TAG=${1:-t03}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 900 python tools/mb.py 1000000 4000000 > $OUT/mb.log 2>&1; echo "mb rc=$?"; grep -v "^{" $OUT/mb.log | grep "theta0\|N  \|scan<8,2,8\|scan<16,1,8\|scan<12,2" | tail -70
for v in 0 1; do
  echo "== bench 32M variant $v"; ATLAS_SCAN_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_32m_v$v.json 2> $OUT/bench_32m_v$v.err; echo "rc=$?"; cat $OUT/bench_32m_v$v.json; tail -3 $OUT/bench_32m_v$v.err
  echo "== bench 4M variant $v"; ATLAS_SCAN_VARIANT=$v timeout 600 python bench.py --passages 4000000 --steps 30 --warmup 3 --cpu-seconds 0 > $OUT/bench_4m_v$v.json 2> $OUT/bench_4m_v$v.err; echo "rc=$?"; cat $OUT/bench_4m_v$v.json
  echo "== bench 1M variant $v"; ATLAS_SCAN_VARIANT=$v timeout 600 python bench.py --passages 1000000 --steps 50 --warmup 5 --cpu-seconds 0 > $OUT/bench_1m_v$v.json 2> $OUT/bench_1m_v$v.err; echo "rc=$?"; cat $OUT/bench_1m_v$v.json
done
echo "== rocprofv3 kernel trace 1M"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_1m -o trace -- python $GRAFT_REPO_ROOT/bench.py --passages 1000000 --steps 30 --warmup 3 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/prof_1m.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof_1m -name "*kernel_stats*.csv" | head -1); do head -14 $f | cut -c1-200; done
