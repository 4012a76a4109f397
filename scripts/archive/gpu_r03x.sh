#!/bin/bash
OUT=gpurun_out/r03x; mkdir -p $OUT
timeout 900 python tools/lib_ab.py scan head=tools/ab/head.so head16=tools/ab/head16.so 5 2>&1 | grep -v amdgpu.ids | grep f32 | tee $OUT/scan_builds.txt
