#!/bin/bash
# round 2: the run-time tile pool of the scan: parity of its configurations, then A/B of the pool share at 1M / 4M / 32M rows
OUT=gpurun_out/r02q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_search.py -m gpu -q --no-header -x -p no:cacheprovider -k "pool or 1m or bit_exact or golden or duplicated" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
grep -E "passed|failed|FAILED|ERROR|Error" $OUT/pytest.log | tail -8 | tee -a $OUT/summary.log
timeout 600 python tools/scan_pool_ab.py ${SIZES:-1000000 4000000 32000000} > $OUT/scan_pool_ab.txt 2>&1; echo "ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/scan_pool_ab.txt | tail -50
