#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout 600 python tools/scan_wg_times.py 1000000 4000000 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_wg_times.txt
