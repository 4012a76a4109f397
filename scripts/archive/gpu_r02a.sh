#!/bin/bash
# round 2, session A: GPU tests (incl. the new 1M / 32M parity tests) + the default bench line
OUT=gpurun_out/r02a; mkdir -p $OUT; export TMPDIR=/tmp
(python -c "import torch;print(torch.__version__, torch.cuda.device_count(), torch.cuda.get_device_name(0))"; nproc; free -g | head -2) > $OUT/env.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee $OUT/summary.log
timeout 1500 python -m pytest tests -m gpu -q --no-header -rA -p no:cacheprovider --durations=15 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.log
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -15 | tee -a $OUT/summary.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" | tee -a $OUT/summary.log
tail -5 $OUT/bench_default.err | tee -a $OUT/summary.log
cat $OUT/bench_default.json | tee -a $OUT/summary.log
