#!/bin/bash
OUT=gpurun_out/r03l; mkdir -p $OUT
timeout 900 python tools/lib_ab.py scan head=tools/ab/head.so park=tools/ab/park.so nopark_r32=tools/ab/nopark_r32.so 4 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_builds2.txt
