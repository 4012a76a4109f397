#!/bin/bash
OUT=gpurun_out/${1:-e03}; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in 2; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/cfg$cfg -o t -- python $GRAFT_REPO_ROOT/bench.py --passages 1000000 --steps 3 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/cfg$cfg.log 2>&1)
  python - <<PY
import csv, json
lines=[l for l in open("$OUT/cfg$cfg.log").read().splitlines() if l.startswith("{")]
d=json.loads(lines[-1])["refresh"]
print("cfg$cfg refresh", round(d["value"]), "passages/s", round(d["roofline"]["achieved"],1), "TF")
for r in csv.DictReader(open("$OUT/cfg$cfg/t_kernel_stats.csv")):
    n=r["Name"]
    if any(k in n for k in ("gemm","attention","ln_kernel","pool","embed_ln","count_k","pack_k")):
        print("   %-60s calls=%4s avg=%9.1f us"%(n[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
timeout 1200 python -m pytest tests/test_gpu_encoder.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | tail -3
for a in "fp16 trim" "fp32 trim" "bf16 trim"; do timeout 300 python tools/enc_time.py $a 2>&1 | grep -v amdgpu.ids | tee -a $OUT/enc_time.log; done
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
