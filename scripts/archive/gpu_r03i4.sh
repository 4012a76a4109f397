#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout 600 python tools/fused_timeline.py 1000000 4000000 2>&1 | grep -v amdgpu.ids | tee $OUT/fused_timeline.txt
