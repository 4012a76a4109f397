#!/bin/bash
# GEMM tuning session: encoder parity tests (optional), kernel trace of tools/gemm_diag.py
OUT=gpurun_out/${1:-g01}; mkdir -p $OUT; export TMPDIR=/tmp
MODES=${MODES:-0 1 2}
if [ "${TESTS:-1}" = "1" ]; then timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_encoder_golden.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | tail -3; fi
for nb in ${NBS:-512}; do
  echo "== NB=$nb passages x 128 tokens" | tee -a $OUT/gemm_diag.txt
  (cd /tmp && NB=$nb ATLAS_GEMM_CFG=${CFG:-4} rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/gemm_diag.py $MODES > $GRAFT_REPO_ROOT/$OUT/diag.log 2>&1)
  python tools/gemm_diag_report.py $(ls $OUT/prof/*kernel_trace.csv | head -1) $MODES | tee -a $OUT/gemm_diag.txt
  rm -rf $OUT/prof
done
