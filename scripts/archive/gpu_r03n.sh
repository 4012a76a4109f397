#!/bin/bash
# (runs on commit c6a8d8f only: the proxy variants of the scan and tools/scan_q96_proxy.py were replaced by the real 96-query pass)
OUT=gpurun_out/r03n; mkdir -p $OUT
timeout 900 python tools/scan_q96_proxy.py 4000000 32000000 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_96_query_proxy.txt
