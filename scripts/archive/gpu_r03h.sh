#!/bin/bash
# round 3, session h: same-process A/B of library BUILDS (tools/lib_ab.py): the encoder before / after the split group-A epilogue and its two
# parts apart; the scan with its cold list / granule addresses formed in place (zero scratch) against the committed one
OUT=gpurun_out/r03h; mkdir -p $OUT
timeout 900 python tools/lib_ab.py enc prev=tools/ab/prev.so head=tools/ab/head.so unsplit2=tools/ab/unsplit2.so lane_resid=tools/ab/lane_resid.so 6 2>&1 | grep -v amdgpu.ids | tee $OUT/enc_builds.txt
timeout 900 python tools/lib_ab.py scan head=tools/ab/head.so scan0=tools/ab/scan0.so 5 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_builds.txt
