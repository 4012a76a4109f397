#!/bin/bash
# round 3, session t: streamed refresh in batches of 65 536 tokens instead of 512 passages -- parity (end-to-end tests), A/B in one process
OUT=gpurun_out/r03t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_encoder.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error" | tail -3 | tee $OUT/summary.log
timeout 900 python tools/streamed_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/streamed_ab.txt
