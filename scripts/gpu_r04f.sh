#!/bin/bash
# round 4, session f: the whole search suite on the GEMM-shaped passes, then the A/B at 32M rows (k = 40) and at 4M rows with k = 100
OUT=gpurun_out/r04f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_search.py tests/test_gpu_parity_32m.py tests/test_gpu_end_to_end.py -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -5 $OUT/pytest_search.log | tee -a $OUT/summary.log
timeout 900 python tools/batch_gemm_ab.py 32000000 > $OUT/batch_gemm_ab_32m.txt 2>&1; echo "ab rc=$?" | tee -a $OUT/summary.log
cut -c1-330 $OUT/batch_gemm_ab_32m.txt | tee -a $OUT/summary.log
timeout 900 python tools/batch_gemm_ab.py 4000000 --k 100 > $OUT/batch_gemm_ab_4m_k100.txt 2>&1; echo "ab rc=$?" | tee -a $OUT/summary.log
cut -c1-330 $OUT/batch_gemm_ab_4m_k100.txt | tee -a $OUT/summary.log
