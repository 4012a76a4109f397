#!/bin/bash
# round 4, session c: phase stamps of the GEMM-shaped scan
OUT=gpurun_out/r04c
mkdir -p $OUT
timeout 300 python tools/gscan_phases.py 4000000 512 > $OUT/gscan_phases_512.txt 2>&1; echo "rc=$?" | tee $OUT/summary.log
timeout 300 python tools/gscan_phases.py 4000000 256 > $OUT/gscan_phases_256.txt 2>&1; echo "rc=$?" | tee -a $OUT/summary.log
head -40 $OUT/gscan_phases_512.txt; tail -3 $OUT/gscan_phases_512.txt; tail -3 $OUT/gscan_phases_256.txt
