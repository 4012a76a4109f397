#!/bin/bash
# One GPU-box session that reproduces what the driver runs at round end + the profiles committed under profiles/.
# usage: scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(python -c "import torch;print(torch.__version__, torch.cuda.device_count(), torch.cuda.get_device_name(0))"; nproc; free -g | head -2) > $OUT/env.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee $OUT/summary.log
timeout 1500 python -m pytest tests -m gpu -q --no-header -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3 | tee -a $OUT/summary.log
echo "== default bench (32M, N=1)" | tee -a $OUT/summary.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" | tee -a $OUT/summary.log
cat $OUT/bench_default.json | tee -a $OUT/summary.log
echo "== rocprofv3 --kernel-trace --stats of the same command (fewer steps)" | tee -a $OUT/summary.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/prof_default.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.log
grep -i "atlas::\|merge_rescore\|prep_queries\|gemm_\|attention_\|ln_kernel\|pool_\|embed_ln\|Name" $OUT/prof_default/trace_kernel_stats.csv | cut -c1-170 | tee -a $OUT/summary.log
for n in 4000000 1000000; do
  echo "== bench $n" | tee -a $OUT/summary.log
  timeout 600 python bench.py --passages $n --steps 50 --warmup 5 --cpu-seconds 0 --refresh-batches 0 > $OUT/bench_$n.json 2> $OUT/bench_$n.err; cat $OUT/bench_$n.json | cut -c1-900 | tee -a $OUT/summary.log
done
