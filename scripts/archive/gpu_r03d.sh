#!/bin/bash
# round 3, session d: the persistent GEMM with the wave-private LDS-transposed epilogue (4 rows x 256 B stores), LDS-DMA bias and residual
# touch: parity, stamps, same-process A/B (tuning build) and the refresh legs of the product build
OUT=gpurun_out/r03d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 tools/hazard_probe > $OUT/hazard_probe.txt 2>&1; cat $OUT/hazard_probe.txt | tee $OUT/summary.log
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_encoder_golden.py -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_encoder.log 2>&1; echo "pytest encoder rc=$?" | tee -a $OUT/summary.log
tail -3 $OUT/pytest_encoder.log | tee -a $OUT/summary.log
timeout 300 python tools/pt_stamps.py > $OUT/pt_stamps.txt 2>&1; echo "pt_stamps rc=$?" | tee -a $OUT/summary.log
cat $OUT/pt_stamps.txt | tee -a $OUT/summary.log
timeout 600 python tools/enc_ab.py 4:0,9:0,4:1,9:1 5 > $OUT/enc_ab.txt 2>&1; echo "enc_ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/enc_ab.txt | tee -a $OUT/summary.log
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --shard-sweep '' --batch-sweep '' --refresh-stream-seconds 3 --passages 4000000 > $OUT/bench_refresh.json 2> $OUT/bench_refresh.err; echo "bench rc=$?" | tee -a $OUT/summary.log
python - <<'PY' | tee -a $OUT/summary.log
import json
d = json.loads(open("gpurun_out/r03d/bench_refresh.json").read().strip().splitlines()[-1])
r = d["refresh"]
print("refresh %.0f passages/s (%.3f ms per batch, frac %.4f)  ragged %.0f  streamed %.0f" % (r["value"], r["ms_per_batch"], r["roofline"]["frac"], r["ragged"]["value"], r["streamed"]["value"]))
PY
