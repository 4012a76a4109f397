#!/bin/bash
# round 4, session h: host-side cost of the synchronous product calls after the pinned-D2H / doc-array change; a short bench line
OUT=gpurun_out/r04h
mkdir -p $OUT
timeout 600 python tools/host_overhead.py 1000000 4000000 > $OUT/host_overhead.txt 2>&1; echo "rc=$?" | tee $OUT/summary.log
grep "^N=" $OUT/host_overhead.txt | tee -a $OUT/summary.log
timeout 900 python bench.py --passages 8000000 --steps 20 --warmup 3 --cpu-seconds 0 --refresh-batches 0 --shard-sweep 1000000,4000000 > $OUT/bench_8m.json 2> $OUT/bench_8m.err; echo "bench rc=$?" | tee -a $OUT/summary.log
python - <<'PY' | tee -a $OUT/summary.log
import json
d = json.loads(open("gpurun_out/r04h/bench_8m.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
for n, v in d["shard_sweep"].items(): print(n, {k: (round(x, 4) if isinstance(x, float) else x) for k, x in v.items() if k not in ("timing", "parity_checked")})
for n, v in d["batch_sweep"].items(): print(n, {k: (round(x, 4) if isinstance(x, float) else x) for k, x in v.items() if k not in ("parity_checked",)})
print(d["detail"])
PY
tail -3 $OUT/bench_8m.err
