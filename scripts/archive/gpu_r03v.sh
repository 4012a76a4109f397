#!/bin/bash
# round 3, session v: bench.py --gpus 2 of the final code under torch.distributed.run over gloo, both ranks on the one GPU (logic check of the N > 1 step, not a measurement)
OUT=gpurun_out/r03v; mkdir -p $OUT
ATLAS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --passages 4000003 --cpu-seconds 0 --refresh-batches 4 --refresh-stream-seconds 2 > $OUT/bench_w2.json 2> $OUT/bench_w2.err; echo "rc=$?" | tee $OUT/summary.log
tail -3 $OUT/bench_w2.err | cut -c1-300 | tee -a $OUT/summary.log
cut -c1-1500 $OUT/bench_w2.json | tee -a $OUT/summary.log
