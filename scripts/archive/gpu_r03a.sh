#!/bin/bash
# round 3, session a: the persistent refresh GEMM (cfg 9) -- parity (encoder tests incl. the bit-for-bit configuration test), same-process
# A/B against the launch-per-tile kernels (cfg 4), per-kernel times under rocprofv3, then the whole -m gpu suite
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
(python -c "import torch;print(torch.__version__, torch.cuda.device_count(), torch.cuda.get_device_name(0))"; nproc) > $OUT/env.log 2>&1
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_encoder_golden.py -m gpu -x -q --no-header -rA -p no:cacheprovider > $OUT/pytest_encoder.log 2>&1; echo "pytest encoder rc=$?" | tee $OUT/summary.log
grep -E "passed|failed|Error|differs" $OUT/pytest_encoder.log | tail -5 | tee -a $OUT/summary.log
timeout 600 python tools/enc_ab.py 4,9 6 0,1 > $OUT/enc_ab.txt 2>&1; echo "enc_ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/enc_ab.txt | tee -a $OUT/summary.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/tools/gemm_diag.py 4:0 9:0 9:1 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.log
grep -i "gemm_\|attention_\|ln_kernel\|pool_\|embed_ln\|Name" $OUT/prof/trace_kernel_stats.csv | cut -c1-200 | tee -a $OUT/summary.log
cp $OUT/prof/trace_kernel_stats.csv $OUT/gemm_cfg4_cfg9_kernel_stats.csv 2>/dev/null
python - <<'PY' | tee -a $OUT/summary.log
# per launch medians in order (3 configurations x 4 embeds x 12 layers) from the kernel trace
import csv, glob, collections, statistics
f = glob.glob("gpurun_out/r03a/prof/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "gemm_" in n or "attention" in n or "ln_kernel" in n:
            d[n.split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in d.items():
        print(f"{k:72s} n={len(v):4d} median {statistics.median(v):8.1f} us  min {min(v):8.1f}")
PY
rm -rf $OUT/prof
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?" | tee -a $OUT/summary.log
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.log
