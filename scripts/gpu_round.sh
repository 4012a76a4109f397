#!/bin/bash
# One GPU-box session: parity tests, bench lines, rocprofv3 kernel trace. Everything lands in gpurun_out/.
# usage: scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env" > $OUT/env.log
(rocm-smi --showproductname 2>/dev/null | head -20; python -c "import torch;print(torch.__version__, torch.cuda.device_count(), torch.cuda.get_device_name(0))"; nproc; free -g | head -2) >> $OUT/env.log 2>&1
echo "== smoke" | tee -a $OUT/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.log
echo "== pytest gpu" | tee -a $OUT/summary.log
timeout 1500 python -m pytest tests -m gpu -q --no-header -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.log
tail -40 $OUT/pytest_gpu.log | grep -E "passed|failed|PASSED|FAILED|ERROR" | tail -40 | tee -a $OUT/summary.log
echo "== bench 1M" | tee -a $OUT/summary.log
timeout 600 python bench.py --passages 1000000 --steps 50 --warmup 5 --cpu-seconds 0 > $OUT/bench_1m.json 2> $OUT/bench_1m.err; echo "rc=$?" | tee -a $OUT/summary.log
cat $OUT/bench_1m.json | tee -a $OUT/summary.log
echo "== bench 32M (default workload)" | tee -a $OUT/summary.log
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_32m.json 2> $OUT/bench_32m.err; echo "rc=$?" | tee -a $OUT/summary.log
cat $OUT/bench_32m.json | tee -a $OUT/summary.log
echo "== rocprofv3 kernel trace (4M shard = 32M/8)" | tee -a $OUT/summary.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_4m -o trace -- python $GRAFT_REPO_ROOT/bench.py --passages 4000000 --steps 20 --warmup 3 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/prof_4m.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.log
find $OUT/prof_4m -name "*stats*" | head | tee -a $OUT/summary.log
for f in $(find $OUT/prof_4m -name "*kernel_stats*.csv" | head -1); do head -12 $f | tee -a $OUT/summary.log; done
tail -5 $OUT/bench_1m.err $OUT/bench_32m.err 2>/dev/null | tail -20
