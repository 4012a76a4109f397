#!/bin/bash
# round 3, session e: persistent GEMM with both groups' epilogues side by side (group A takes the next tile's first barrier early), the
# merge emitting packed candidates, the outlier-weights encoder tests: whole -m gpu suite, stamps, A/B, refresh legs of the product build
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?" | tee $OUT/summary.log
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -8 | tee -a $OUT/summary.log
grep -E "pooling=|real checkpoint" $OUT/pytest_gpu.log | tee -a $OUT/summary.log
timeout 300 python tools/pt_stamps.py > $OUT/pt_stamps.txt 2>&1; echo "pt_stamps rc=$?" | tee -a $OUT/summary.log
cat $OUT/pt_stamps.txt | tee -a $OUT/summary.log
timeout 600 python tools/enc_ab.py 4:0,9:0,4:1,9:1 5 > $OUT/enc_ab.txt 2>&1; echo "enc_ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/enc_ab.txt | tee -a $OUT/summary.log
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --shard-sweep '' --batch-sweep '' --refresh-stream-seconds 3 --passages 4000000 > $OUT/bench_refresh.json 2> $OUT/bench_refresh.err; echo "bench rc=$?" | tee -a $OUT/summary.log
python - <<'PY' | tee -a $OUT/summary.log
import json
d = json.loads(open("gpurun_out/r03e/bench_refresh.json").read().strip().splitlines()[-1])
r = d["refresh"]
print("refresh %.0f passages/s (%.3f ms per batch, frac %.4f)  ragged %.0f  streamed %.0f" % (r["value"], r["ms_per_batch"], r["roofline"]["frac"], r["ragged"]["value"], r["streamed"]["value"]))
PY
