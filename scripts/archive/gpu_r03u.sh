#!/bin/bash
# round 3, session u: per-layer GEMM times of the full refresh batch, persistent kernel (cfg 9) vs the round-2 kernels (cfg 4), with and without epilogues
OUT=gpurun_out/r03u; mkdir -p $OUT; export TMPDIR=/tmp
MODES="9:0 4:0 9:1 4:1"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/gemm_diag.py $MODES > $GRAFT_REPO_ROOT/$OUT/diag.log 2>&1); echo "diag rc=$?"
python tools/gemm_layer_report.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) $MODES | tee $OUT/gemm_layer_report.txt
rm -rf $OUT/prof
