#!/bin/bash
# round 4, session d: after a change of gscan_kernel.h: big-batch parity tests, A/B at 4M rows, phase stamps
OUT=gpurun_out/r04d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -q --no-header -x -p no:cacheprovider -k "batches_above or certifying" > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -3 $OUT/pytest_search.log | tee -a $OUT/summary.log
timeout 900 python tools/batch_gemm_ab.py ${ROWS:-4000000} > $OUT/batch_gemm_ab.txt 2>&1; echo "ab rc=$?" | tee -a $OUT/summary.log
cut -c1-330 $OUT/batch_gemm_ab.txt | tee -a $OUT/summary.log
timeout 300 python tools/gscan_phases.py 4000000 512 > $OUT/gscan_phases_512.txt 2>&1
grep -A 14 "wave 0" $OUT/gscan_phases_512.txt | head -16; grep -A 14 "wave 4" $OUT/gscan_phases_512.txt | head -16; tail -1 $OUT/gscan_phases_512.txt
