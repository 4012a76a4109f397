#!/bin/bash
# round 2, session D: search-path GPU tests + search-only bench + per-workgroup stamps after a scan change
OUT=gpurun_out/r02d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_parity_32m.py -m gpu -q --no-header -x -p no:cacheprovider > $OUT/pytest_search.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest_search.log
timeout 600 python bench.py --refresh-batches 0 --cpu-seconds 0 > $OUT/bench_search.json 2> $OUT/bench_search.err; echo "bench rc=$?"; tail -3 $OUT/bench_search.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_search.json").read().strip().splitlines()[-1])
print("32M: %.1f q/s step %.3f ms kernel %.3f ms frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_mean"], d["roofline"]["frac"]))
for n, v in d["shard_sweep"].items(): print(n, {k: round(x, 4) for k, x in v.items()})
print(d["detail"])
PY
timeout 300 python tools/merge_phases.py 1000000 4000000 > $OUT/merge_phases.txt 2>&1; tail -2 $OUT/merge_phases.txt; timeout 300 python tools/scan_wg_times.py 1000000 4000000 > $OUT/scan_wg_times.txt 2>&1; grep -E "kernel span|WGs still|mean loop per XCD" $OUT/scan_wg_times.txt
