#!/bin/bash
# round 3, session b: what bounds the GEMM epilogues (store-burst microbenchmark; store policies and a start stagger in the persistent
# kernel), the shifted V^T layout (encoder parity), and the trusted-pmax nets (search tests)
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 tools/store_bench > $OUT/store_bench.txt 2>&1; echo "store_bench rc=$?" | tee $OUT/summary.log
cat $OUT/store_bench.txt | tee -a $OUT/summary.log
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_encoder_golden.py -m gpu -x -q --no-header -p no:cacheprovider > $OUT/pytest_encoder.log 2>&1; echo "pytest encoder rc=$?" | tee -a $OUT/summary.log
tail -3 $OUT/pytest_encoder.log | tee -a $OUT/summary.log
timeout 600 python -m pytest tests/test_gpu_search.py -m gpu -x -q --no-header -p no:cacheprovider -k "version_counter or large_norm or certifying or exact_path or bit_exact" > $OUT/pytest_search.log 2>&1; echo "pytest search rc=$?" | tee -a $OUT/summary.log
tail -3 $OUT/pytest_search.log | tee -a $OUT/summary.log
timeout 900 python tools/enc_ab.py 4:0,9:0,9:4,9:8,9:5632,9:11264,4:1,9:1 5 > $OUT/enc_ab.txt 2>&1; echo "enc_ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/enc_ab.txt | tee -a $OUT/summary.log
