#!/bin/bash
# round 3, session y: the peer exchange with two processes on the one GPU
OUT=gpurun_out/r03y; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_peer_exchange.py -m gpu -q --no-header -x -p no:cacheprovider -rs 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 | tee $OUT/summary.log
