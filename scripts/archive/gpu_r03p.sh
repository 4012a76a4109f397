#!/bin/bash
# round 3, session p (final state of the round): what the driver runs at round end + the profiles committed under profiles/r03/:
# smoke, the -m gpu suite, the default bench line, rocprofv3 --kernel-trace --stats of the driver-shaped command, PMC FETCH_SIZE passes
# (own runs) at 1M / 4M / 32M rows, MFMA-busy PMC passes of the refresh encoder, tile-boundary stamps and the same-process A/B
OUT=gpurun_out/r03p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee $OUT/summary.log
timeout 900 python -m pytest tests -m gpu -q --no-header -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.log
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -5 | tee -a $OUT/summary.log
grep -E "synchronous search at 2M" $OUT/pytest_gpu.log | tee -a $OUT/summary.log
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" | tee -a $OUT/summary.log
cut -c1-1500 $OUT/bench_default.json | tee -a $OUT/summary.log
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_default -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --refresh-stream-seconds 3 > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/prof_default.err); echo "rocprof rc=$?" | tee -a $OUT/summary.log
f=$(find $OUT/prof_default -name "*kernel_stats*.csv" | head -1); cp $f $OUT/bench_default_kernel_stats.csv; grep -i "scan_kernel\|merge_rescore\|gemm_\|attention_\|ln_kernel\|pool_\|embed_ln\|Name" $OUT/bench_default_kernel_stats.csv | cut -c1-40,140-330 | tee -a $OUT/summary.log
rm -rf $OUT/prof_default
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w tools/microbench.hip -o tools/libatlas_mb.so 2>&1 | tail -2
for n in 1000000 4000000 32000000; do
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $n > $GRAFT_REPO_ROOT/$OUT/pmc_$n.log 2>&1); echo "pmc $n rc=$?" | tee -a $OUT/summary.log
  cp $(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1) $OUT/pmc_${n}_fetch_counter_collection.csv
done
python tools/pmc_summarize.py $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000 | tee -a $OUT/summary.log
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
rm -rf $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/enc -o pmc -- python $GRAFT_REPO_ROOT/tools/enc_pmc_run.py 2 > $GRAFT_REPO_ROOT/$OUT/enc_pmc.log 2>&1); echo "encoder pmc rc=$?" | tee -a $OUT/summary.log
python tools/pmc_mfma_summarize.py $OUT/enc | tee $OUT/mfma_util.txt | tee -a $OUT/summary.log
rm -rf $OUT/enc
timeout 300 python tools/pt_stamps.py > $OUT/pt_stamps.txt 2>&1; echo "pt_stamps rc=$?" | tee -a $OUT/summary.log
grep "wave 0 tile 1" $OUT/pt_stamps.txt | cut -c1-420 | tee -a $OUT/summary.log
timeout 600 python tools/enc_ab.py 4:0,9:0,4:1,9:1 5 > $OUT/enc_ab.txt 2>&1; echo "enc_ab rc=$?" | tee -a $OUT/summary.log
cat $OUT/enc_ab.txt | tee -a $OUT/summary.log
