#!/bin/bash
# cache-policy variants of the scan kernel + the N=2 logic check of bench.py (gloo, two ranks on the one GPU)
OUT=gpurun_out/${1:-p01}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/scan_policy.py 1000000 4000000 32000000 -- 0 5 6 2>&1 | grep -v amdgpu.ids | tee $OUT/scan_policy.txt
echo "== bench.py --gpus 2 over gloo on one GPU (logic check of the N>1 step: pack -> all-gather -> merge)" | tee $OUT/bench_w2_gloo.txt
ATLAS_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
   bench.py --gpus 2 --steps 10 --warmup 2 --passages 4000000 --refresh-batches 1 2>&1 | grep -v amdgpu.ids | tail -5 | tee -a $OUT/bench_w2_gloo.txt
