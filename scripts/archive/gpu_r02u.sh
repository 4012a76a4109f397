#!/bin/bash
# round 2: MFMA-pipe utilisation of the dominant kernels from PMC counters (own passes, --kernel-trace only): the scan at 32M rows and
# the refresh encoder (2 layers of the 512 x 128-token fp16 batch)
OUT=gpurun_out/r02u; mkdir -p $OUT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w tools/microbench.hip -o tools/libatlas_mb.so 2>&1 | tail -2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/scan -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py ${ROWS:-32000000} > $GRAFT_REPO_ROOT/$OUT/scan.log 2>&1); echo "scan pmc rc=$?"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/enc -o pmc -- python $GRAFT_REPO_ROOT/tools/enc_pmc_run.py 2 > $GRAFT_REPO_ROOT/$OUT/enc.log 2>&1); echo "encoder pmc rc=$?"
python tools/pmc_mfma_summarize.py $OUT/scan $OUT/enc | tee $OUT/mfma_util.txt
rm -rf $OUT/scan $OUT/enc
