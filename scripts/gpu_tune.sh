#!/bin/bash
TAG=${1:-t10}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
python tools/merge_phases.py 1000000 4000000 2>&1 | grep -v amdgpu.ids
for n in 1000000 4000000; do
echo "== rocprofv3 kernel trace $n"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$n -o trace -- python $GRAFT_REPO_ROOT/bench.py --passages $n --steps 30 --warmup 3 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/prof_$n.log 2>&1); echo "rocprof rc=$?"
grep '"metric"' $OUT/prof_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['frac'])"
for f in $(find $OUT/prof_$n -name "*kernel_stats*.csv" | head -1); do grep -i "atlas\|merge\|prep\|Name" $f | cut -c1-150; done
done
echo "== bench 1M (no profiler)"; timeout 600 python bench.py --passages 1000000 --steps 50 --warmup 5 --cpu-seconds 0 > $OUT/bench_1m.json 2> $OUT/bench_1m.err; cat $OUT/bench_1m.json
echo "== bench 4M (no profiler)"; timeout 600 python bench.py --passages 4000000 --steps 50 --warmup 5 --cpu-seconds 0 > $OUT/bench_4m.json 2> $OUT/bench_4m.err; cat $OUT/bench_4m.json
