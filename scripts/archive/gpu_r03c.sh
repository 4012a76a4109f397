#!/bin/bash
# round 3, session c: the store-data hazard probe, encoder parity after the soffset fix, tile-boundary stamps of the persistent GEMM,
# A/B of the configurations, the whole -m gpu suite and the default bench line with its new legs
OUT=gpurun_out/r03c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 tools/hazard_probe > $OUT/hazard_probe.txt 2>&1; echo "hazard_probe rc=$?" | tee $OUT/summary.log
cat $OUT/hazard_probe.txt | tee -a $OUT/summary.log
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_encoder_golden.py -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_encoder.log 2>&1; echo "pytest encoder rc=$?" | tee -a $OUT/summary.log
tail -3 $OUT/pytest_encoder.log | tee -a $OUT/summary.log
timeout 300 python tools/pt_stamps.py > $OUT/pt_stamps.txt 2>&1; echo "pt_stamps rc=$?" | tee -a $OUT/summary.log
cat $OUT/pt_stamps.txt | tee -a $OUT/summary.log
timeout 600 python tools/enc_ab.py 4:0,9:0,4:1,9:1 5 > $OUT/enc_ab.txt 2>&1; echo "enc_ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/enc_ab.txt | tee -a $OUT/summary.log
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?" | tee -a $OUT/summary.log
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3 | tee -a $OUT/summary.log
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" | tee -a $OUT/summary.log
tail -3 $OUT/bench_default.err | tee -a $OUT/summary.log
cut -c1-3000 $OUT/bench_default.json | tee -a $OUT/summary.log
