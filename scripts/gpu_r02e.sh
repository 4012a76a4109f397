#!/bin/bash
OUT=gpurun_out/r02e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/scan_wg_times.py 1000000 > $OUT/scan_wg_times.txt 2>&1; grep -E "kernel span|image barrier|published|collected|query image|first tile" $OUT/scan_wg_times.txt | head -14
