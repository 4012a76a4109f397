#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout 900 python tools/lib_ab.py scan head=tools/ab/head.so fused=tools/ab/fused.so 5 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_builds.txt
