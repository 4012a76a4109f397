#!/bin/bash
OUT=gpurun_out/r02e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_parity_32m.py -m gpu -q --no-header -x -p no:cacheprovider > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_search.log | tail -1
timeout 600 python tools/scan_ab.py 1000000 4000000 > $OUT/scan_ab.txt 2>&1; grep "^N=" $OUT/scan_ab.txt
timeout 300 python tools/scan_wg_times.py 1000000 > $OUT/scan_wg_times.txt 2>&1; grep -E "kernel span|image barrier|published|collected|query image|first tile" $OUT/scan_wg_times.txt | head -7
