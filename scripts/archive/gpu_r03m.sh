#!/bin/bash
OUT=gpurun_out/r03m; mkdir -p $OUT
timeout 900 python tools/lib_ab.py enc prev=tools/ab/prev.so head=tools/ab/head.so unsplit4=tools/ab/unsplit4.so unsplit24=tools/ab/unsplit24.so 6 2>&1 | grep -v amdgpu.ids | tee $OUT/enc_builds.txt
