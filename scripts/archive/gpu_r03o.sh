#!/bin/bash
# round 3, session o: the 96-query slab pass -- parity of the whole search suite (batches above 64 queries included), batch sweep A/B
OUT=gpurun_out/r03o; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_parity_32m.py tests/test_gpu_end_to_end.py -m gpu -q --no-header -x -p no:cacheprovider > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -5 $OUT/pytest_search.log | cut -c1-300 | tee -a $OUT/summary.log
timeout 900 python tools/batch_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/batch_ab.txt
