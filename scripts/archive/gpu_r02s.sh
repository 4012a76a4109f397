#!/bin/bash
# what the box's clocks / power look like while the search bench runs (boxes differ by 5-8 %: which knob is it?)
OUT=gpurun_out/r02s; mkdir -p $OUT; export TMPDIR=/tmp
rocm-smi --showperflevel --showclocks --showpower --showtemp --showmemvendor --showvoltage > $OUT/smi_idle.txt 2>&1
rocm-smi --showmclkrange --showsclkrange > $OUT/smi_ranges.txt 2>&1
( for i in $(seq 1 60); do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)"; sleep 0.4; done ) > $OUT/smi_during.txt 2>&1 &
SMI=$!
timeout 600 python bench.py --steps 200 --warmup 5 --refresh-batches 0 --cpu-seconds 0 --shard-sweep 4000000 > $OUT/bench_search.json 2> $OUT/bench_search.err; echo "bench rc=$?"
kill $SMI 2>/dev/null
python - <<'PY'
import json,re
d=json.load(open("gpurun_out/r02s/bench_search.json"))
print("32M kernel frac", round(d["roofline"]["frac"],4), "kernel ms", round(d["roofline"]["kernel_ms_mean"],4), "4M step frac", round(d["shard_sweep"]["4000000"]["step_frac"],4))
PY
grep -E "mclk|fclk|sclk|Power" $OUT/smi_during.txt | sort | uniq -c | sort -rn | head -16
