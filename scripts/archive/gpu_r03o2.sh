#!/bin/bash
OUT=gpurun_out/r03o; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_search.py tests/test_gpu_parity_32m.py tests/test_gpu_end_to_end.py -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -6 $OUT/pytest_search.log | cut -c1-300 | tee -a $OUT/summary.log
