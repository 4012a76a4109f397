#!/bin/bash
# round 4, session j: refresh GEMM with the LDS-DMA pieces split 11 + 5 between the ping-pong groups: encoder parity tests, then same-process A/B of the builds
OUT=gpurun_out/r04j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_encoder_outliers.py tests/test_encoder_golden.py -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest_encoder.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.log
tail -3 $OUT/pytest_encoder.log | tee -a $OUT/summary.log
timeout 900 python tools/lib_ab.py enc $(for f in tools/abx/*.so; do echo -n "$(basename $f .so)=$f "; done) 7 > $OUT/enc_ab.txt 2>&1; echo "ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/enc_ab.txt | tee -a $OUT/summary.log
