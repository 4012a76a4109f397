#!/bin/bash
# round 3, session i: the merge fused into the scan's last workgroups -- parity suite of the search path, then same-process A/B of the builds
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_parity_32m.py tests/test_gpu_end_to_end.py -m gpu -q --no-header -x -p no:cacheprovider > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -5 $OUT/pytest_search.log | tee -a $OUT/summary.log
grep -E "synchronous search at 2M" $OUT/pytest_search.log | tee -a $OUT/summary.log
timeout 900 python tools/lib_ab.py scan head=tools/ab/head.so fused=tools/ab/fused.so 5 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_builds.txt
