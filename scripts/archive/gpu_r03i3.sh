#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout 900 python tools/lib_ab.py scan head=tools/ab/head.so fused=tools/ab/fused.so fused_off=tools/ab/fused_off.so norel=tools/ab/fused_norel.so noacq=tools/ab/fused_noacq.so nofence=tools/ab/fused_nofence.so 5 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_builds_exp.txt
