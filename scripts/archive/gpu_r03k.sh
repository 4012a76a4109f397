#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 900 python tools/lib_ab.py scan head=tools/ab/head.so imgfirst=tools/ab/imgfirst.so imgrot=tools/ab/imgrot.so imgbar=tools/ab/imgbar.so 5 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_builds.txt
timeout 600 python tools/scan_wg_times.py 1000000 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_wg_times.txt
