#!/bin/bash
# round 4, session i: the reference's own Atlas class on the MI355X with both HIP back-ends (tests/test_gpu_reference_atlas.py; the reference files ride along in .refstage/)
OUT=gpurun_out/r04i
mkdir -p $OUT
ls .refstage/src > $OUT/refstage.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_reference_atlas.py -m gpu -q --no-header -p no:cacheprovider -rA -s > $OUT/pytest_reference_atlas_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
grep -E "reference Atlas|passed|failed|skipped|Error" $OUT/pytest_reference_atlas_gpu.log | tee -a $OUT/summary.log
