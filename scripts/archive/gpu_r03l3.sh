#!/bin/bash
OUT=gpurun_out/r03l; mkdir -p $OUT
ATLAS_HIP_SO=$PWD/tools/ab/head_t.so timeout 600 python tools/scan_wg_times.py 4000000 2>&1 | grep -v amdgpu.ids > $OUT/wg_head_4m.txt
ATLAS_HIP_SO=$PWD/tools/ab/park_t.so timeout 600 python tools/scan_wg_times.py 4000000 2>&1 | grep -v amdgpu.ids > $OUT/wg_park_4m.txt
