#!/bin/bash
# round 3, session r: the search suite three times over (flakiness of the in-kernel exchanges / the wide pass), then the whole -m gpu suite once
OUT=gpurun_out/r03r; mkdir -p $OUT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1; done | tee $OUT/stress.log
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $OUT/stress.log
