#!/bin/bash
# round 3: MFMA-pipe utilisation of the scan's 64- and 96-query passes (PMC, own pass) at 32M rows
OUT=gpurun_out/r03aa; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/scan -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run_wide.py ${ROWS:-32000000} > $GRAFT_REPO_ROOT/$OUT/scan.log 2>&1); echo "scan pmc rc=$?"
python tools/pmc_mfma_summarize.py $OUT/scan | tee $OUT/mfma_util_scan.txt
rm -rf $OUT/scan
