#!/bin/bash
# round 4, session g: search suite incl. the GEMM-shaped passes' tests (both twins)
OUT=gpurun_out/r04g
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_search.py -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -15 $OUT/pytest_search.log | tee -a $OUT/summary.log
