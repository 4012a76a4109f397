#!/bin/bash
OUT=gpurun_out/r02m; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_shapes.hip -o /tmp/mfma_shapes 2>&1 | grep -E "error" ; timeout 300 /tmp/mfma_shapes > $OUT/mfma_shapes.txt 2>&1; cat $OUT/mfma_shapes.txt
