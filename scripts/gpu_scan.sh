#!/bin/bash
# scan-kernel session: parity tests, bench at three shard sizes, per-workgroup stamps
OUT=gpurun_out/${1:-s01}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_search.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | tail -3
for n in 1000000 4000000 32000000; do
  timeout 600 python bench.py --passages $n --steps 40 --warmup 5 --cpu-seconds 0 --refresh-batches 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N=%9d  %9.0f q/s  step %.4f ms  scan mean %.4f min %.4f ms  frac %.3f' % ($n, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['kernel_ms_min'], d['roofline']['frac']))"
done
python tools/scan_wg_times.py 4000000 1000000 2>&1 | grep -v amdgpu | grep "N=\|loop\|hand-over\|per XCD"
