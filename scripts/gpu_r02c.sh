#!/bin/bash
# round 2, session C: where does the size-independent part of the scan time go? wall-clock stamps per workgroup + the tail experiment
OUT=gpurun_out/r02c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/scan_tail.py 1000000 4000000 > $OUT/scan_tail.txt 2>&1; echo "tail rc=$?"
true
cat $OUT/scan_tail.txt
