#!/bin/bash
OUT=gpurun_out/r04e
mkdir -p $OUT
timeout 300 python tools/gscan_phases.py 4000000 512 > $OUT/gscan_phases_512.txt 2>&1
grep -A 14 "wave 0" $OUT/gscan_phases_512.txt | head -16; tail -4 $OUT/gscan_phases_512.txt
