#!/bin/bash
# GEMM A/B: bit-equality of the configurations, then per-GEMM kernel times (rocprofv3 kernel trace) for cfg:diag modes
OUT=gpurun_out/r02g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q --no-header -x -p no:cacheprovider -k "bulk_gemm" > $OUT/pytest_gemm.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|differs" $OUT/pytest_gemm.log | tail -4
MODES="${MODES:-7:0 8:0 7:1 8:1}"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/gemm_diag.py $MODES > $GRAFT_REPO_ROOT/$OUT/diag.log 2>&1); echo "diag rc=$?"; tail -2 $OUT/diag.log
python tools/gemm_diag_report.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) $MODES | tee $OUT/gemm_diag_report.txt
rm -rf $OUT/prof
