#!/bin/bash
# round 3, session ab: paired passes (two query chunks concurrently on half the chip each) -- search suite, single-pass regression A/B of the builds, batch A/B
OUT=gpurun_out/r03ab; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_parity_32m.py tests/test_gpu_end_to_end.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee $OUT/summary.log
LIB_AB_ROWS=4000000 timeout 600 python tools/lib_ab.py scan head=tools/ab/head.so pair=tools/ab/pair.so 4 2>&1 | grep "f32" | tee $OUT/scan_builds.txt
timeout 900 python tools/batch_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/batch_ab.txt
