#!/bin/bash
# round 4, session a: the GEMM-shaped scan of big batches (gscan_kernel.h): parity tests of the search suite, then the A/B against round 3's passes
OUT=gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -q --no-header -x -p no:cacheprovider -k "batches_above or scan_bit_exact or certifying or golden" > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -5 $OUT/pytest_search.log | tee -a $OUT/summary.log
timeout 900 python tools/batch_gemm_ab.py 4000000 > $OUT/batch_gemm_ab_4m.txt 2>&1; echo "ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/batch_gemm_ab_4m.txt | tee -a $OUT/summary.log
