#!/bin/bash
# ONE parametrised GPU-box session (round 4 on; the one-off scripts of rounds 1-3 are under scripts/archive/):
#     gpurun -- 'bash scripts/gpu_session.sh <tag> <step> [<step> ...]'
# outputs under gpurun_out/<tag>/, summary in gpurun_out/<tag>/summary.log; what is kept for the record is copied to profiles/rNN/ by hand.
# steps:
#   smoke          __graft_entry__.smoke()
#   pytest         the whole -m gpu suite                     pytest:<expr>   only `-k <expr>` of it
#   bench          the default bench.py line (what the driver runs)
#   stats32m       rocprofv3 --kernel-trace --stats of a 32M-ONLY search bench (no sweeps, no refresh): isolated averages of the two scan twins
#   stats          rocprofv3 --kernel-trace --stats of the driver-shaped command (all legs)
#   pmc_fetch      PMC FETCH_SIZE passes (own runs) of the 64-query scan at 1M / 4M / 32M rows -> profiles/pmc_traffic.json
#   gscan_ab       tools/batch_gemm_ab.py at 4M and 32M rows (streaming passes vs GEMM-shaped passes, same process)
#   dscan_prof     MFMA-busy and LDS bank-conflict PMC passes of the DMA-staged 64-query scan at 32M rows
#   gscan_prof     per-kernel durations + MFMA-busy + FETCH_SIZE counters of the GEMM-shaped pass (4M rows x 512 / 256 queries), phase stamps
#   enc_pmc        MFMA-busy PMC passes of the refresh encoder (two layers) + per-layer GEMM report
#   gemm_alias     per-layer GEMM times with diag bits: 1 = no epilogue, 16 / 32 = activation / weight loads aliased to the first tile (always L2 hits)
#   pt_cycles      tools/pt_cycles.py: shader cycles per k-tile of the refresh GEMM from end stamps only, per diag mode
#   ldsprobe       tools/lds_read_probe.hip: what a read phase (24 ds_read_b128 per wave, ping-pong, barriers) costs by itself
#   host           tools/host_overhead.py 1M 4M
#   refatlas       tests/test_gpu_reference_atlas.py (needs .refstage/: scripts/stage_reference.sh in the build container)
#   gloo2          two ranks on one GPU over gloo: bench.py --gpus 2 logic check, replicated and --distinct-queries
#   libab_enc / libab_scan   tools/lib_ab.py enc | scan on BUILDS of the library ($LIBAB, default tools/ab/r05.so vs tools/ab/r06.so: scripts/build_variant.sh)
#   enc_knob       tools/enc_knob_ab.py: attention kernels (round 5 | persistent prefetching, 2 / 3 workgroups per CU) and the no-LayerNorm bound, ms + W + mJ per passage
#   encpower       tools/refresh_power.py random zero random zero: the refresh batch with all-zero operands (same instructions, no toggling) beside the real one
#   fullshard      BASELINE configs[3]'s per-GPU share: bench.py --refresh-full-shard 4000000 (one streamed refresh of 4M ragged passages, ~2 min), the
#                  other legs cut short
#   realdry        scripts/real_assets.sh on STAND-IN assets (tools/make_fake_assets.py: random weights, made-up vocabulary / corpus / queries): a dry run
#                  of the real-asset pipeline (encoder tests with ATLAS_CONTRIEVER_DIR set, refresh incl. HF tokenisation, retrieve-only loop)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
say() { echo "$@" | tee -a $OUT/summary.log; }
for STEP in "$@"; do
case $STEP in
realdry)
  python tools/make_fake_assets.py /tmp/fake_assets --layers 12 --passages 30000 --queries 512 > $OUT/fake_assets.log 2>&1
  [ -f tests/golden/enc_real_fake.npz ] && export ATLAS_REAL_GOLDEN=$R/tests/golden/enc_real_fake.npz
  ATLAS_CONTRIEVER_DIR=/tmp/fake_assets/contriever ATLAS_PASSAGES=/tmp/fake_assets/passages.jsonl ATLAS_QUERIES=/tmp/fake_assets/queries.jsonl \
    timeout 1500 bash scripts/real_assets.sh $OUT/real_assets_dry > $OUT/real_assets_dry.log 2>&1; say "realdry rc=$?"; cat $OUT/real_assets_dry/summary.log | cut -c1-700 | tee -a $OUT/summary.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; say "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-400 | tee -a $OUT/summary.log ;;
pytest)
  timeout 1800 python -m pytest tests -m gpu -q --no-header -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; say "pytest rc=$?"
  grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -5 | tee -a $OUT/summary.log ;;
pytest:*)
  timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "${STEP#pytest:}" > $OUT/pytest_k.log 2>&1; say "pytest -k rc=$?"; tail -3 $OUT/pytest_k.log | tee -a $OUT/summary.log ;;
bench)
  timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; say "bench rc=$?"; cut -c1-1200 $OUT/bench_default.json | tee -a $OUT/summary.log ;;
stats32m)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof32 -o trace -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --refresh-batches 0 --shard-sweep '' --batch-sweep '' --emulate-ranks '' > $R/$OUT/bench_32m_only_under_rocprof.json 2> $R/$OUT/prof32.err); say "stats32m rc=$?"
  f=$(find $OUT/prof32 -name "*kernel_stats*.csv" | head -1); cp $f $OUT/bench_32m_only_kernel_stats.csv; grep -i "scan_kernel\|merge_rescore\|slab_pmax\|Name" $OUT/bench_32m_only_kernel_stats.csv | cut -c1-260 | tee -a $OUT/summary.log; rm -rf $OUT/prof32 ;;
stats)
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_default -o trace -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --refresh-stream-seconds 3 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/prof_default.err); say "stats rc=$?"
  f=$(find $OUT/prof_default -name "*kernel_stats*.csv" | head -1); cp $f $OUT/bench_default_kernel_stats.csv; grep -i "scan_kernel\|gscan\|gtheta\|merge_rescore\|gemm_\|attention_\|ln_kernel\|pool_\|embed_ln\|Name" $OUT/bench_default_kernel_stats.csv | cut -c1-40,140-330 | tee -a $OUT/summary.log; rm -rf $OUT/prof_default ;;
pmc_fetch)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w tools/microbench.hip -o tools/libatlas_mb.so 2>&1 | tail -2
  for n in 1000000 4000000 32000000; do
    (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_$n -o pmc -- python $R/tools/pmc_run.py $n > $R/$OUT/pmc_$n.log 2>&1); say "pmc $n rc=$?"
    cp $(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1) $OUT/pmc_${n}_fetch_counter_collection.csv
  done
  python tools/pmc_summarize.py $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000 | tee -a $OUT/summary.log
  cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; rm -rf $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000 ;;
gscan_ab)
  for n in 4000000 32000000; do timeout 900 python tools/batch_gemm_ab.py $n > $OUT/batch_gemm_pass_ab_$n.txt 2>&1; say "gscan_ab $n rc=$?"; cut -c1-330 $OUT/batch_gemm_pass_ab_$n.txt | grep rows | tee -a $OUT/summary.log; done ;;
dscan_prof)
  # MFMA-pipe busy and LDS bank conflicts of the DMA-staged 64-query scan (own PMC passes, --kernel-trace only): tools/pmc_run.py = calibration streams + 4 searches
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w tools/microbench.hip -o tools/libatlas_mb.so 2>&1 | tail -2
  (cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$OUT/dpmc_mfma -o t -- python $R/tools/pmc_run.py 32000000 > $R/$OUT/dpmc_mfma.log 2>&1); say "dscan pmc mfma rc=$?"
  python tools/pmc_mfma_summarize.py $OUT/dpmc_mfma | tee $OUT/mfma_util_dscan.txt | tee -a $OUT/summary.log
  (cd /tmp && timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/dpmc_lds -o t -- python $R/tools/pmc_run.py 32000000 > $R/$OUT/dpmc_lds.log 2>&1); say "dscan pmc lds rc=$?"
  python tools/pmc_lds_summarize.py $OUT/dpmc_lds | tee $OUT/lds_conflicts_dscan.txt | tee -a $OUT/summary.log
  cp $(find $OUT/dpmc_mfma -name "*counter_collection.csv" | head -1) $OUT/dscan_pmc_mfma_counter_collection.csv; cp $(find $OUT/dpmc_lds -name "*counter_collection.csv" | head -1) $OUT/dscan_pmc_lds_counter_collection.csv
  rm -rf $OUT/dpmc_mfma $OUT/dpmc_lds ;;
gscan_prof)
  for B in 256 512; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_$B -o t -- python $R/tools/gscan_run.py 4000000 $B 10 > $R/$OUT/kt_$B.log 2>&1)
    say "== gscan kernel stats, 4000000 rows x $B queries"; grep -E "Name|gscan|gtheta|gprep|merge_rescore" $OUT/kt_$B/t_kernel_stats.csv | cut -c1-200 | tee -a $OUT/summary.log
    cp $OUT/kt_$B/t_kernel_stats.csv $OUT/gscan_4m_${B}q_kernel_stats.csv; rm -rf $OUT/kt_$B
  done
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$OUT/gpmc_mfma -o t -- python $R/tools/gscan_run.py 4000000 512 4 > $R/$OUT/gpmc_mfma.log 2>&1); say "gscan pmc mfma rc=$?"
  python tools/pmc_mfma_summarize.py $OUT/gpmc_mfma | tee $OUT/mfma_util_gscan.txt | tee -a $OUT/summary.log
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/gpmc_fetch -o t -- python $R/tools/gscan_run.py 4000000 512 4 > $R/$OUT/gpmc_fetch.log 2>&1); say "gscan pmc fetch rc=$?"
  python - <<PY | tee -a $OUT/summary.log
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/gpmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gscan_kernel<0" in r["Kernel_Name"] or "gscan_kernelILi0" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in acc.items():
    per_call = sum(v) / (len(v) / 2)          # two scan launches per search
    print("gscan_kernel<0>, 4M rows x 512 queries: FETCH_SIZE %.0f KiB per search (2 launches) = %.3f x the slab's %d bytes after the gfx950 x 2 correction for 16-byte-per-lane reads (uncalibrated for LDS-DMA: an upper bound of ~2 slab reads would be 2.0)" % (per_call, per_call * 1024 * 2 / (4000000 * 1536), 4000000 * 1536))
PY
  cp $(find $OUT/gpmc_mfma -name "*counter_collection.csv" | head -1) $OUT/gscan_pmc_mfma_counter_collection.csv; cp $(find $OUT/gpmc_fetch -name "*counter_collection.csv" | head -1) $OUT/gscan_pmc_fetch_counter_collection.csv
  rm -rf $OUT/gpmc_mfma $OUT/gpmc_fetch
  timeout 300 python tools/gscan_phases.py 4000000 512 > $OUT/gscan_phases_512.txt 2>&1; tail -4 $OUT/gscan_phases_512.txt | cut -c1-200 | tee -a $OUT/summary.log ;;
enc_pmc)
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$OUT/enc -o pmc -- python $R/tools/enc_pmc_run.py 2 > $R/$OUT/enc_pmc.log 2>&1); say "encoder pmc rc=$?"
  python tools/pmc_mfma_summarize.py $OUT/enc | tee $OUT/mfma_util_encoder.txt | tee -a $OUT/summary.log; rm -rf $OUT/enc
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/gdiag -o t -- python $R/tools/gemm_diag.py 9:0 10:0 > $R/$OUT/gemm_diag.log 2>&1); say "gemm_diag rc=$?"
  python tools/gemm_layer_report.py $(find $OUT/gdiag -name "*kernel_trace.csv" | head -1) 9:0 10:0 | tee $OUT/gemm_layer_report.txt | tee -a $OUT/summary.log; rm -rf $OUT/gdiag ;;
gemm_alias)
  M=${GEMM_ALIAS_MODES:-"9:0 9:1 9:17 9:33 9:49 9:16 9:32"}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/galias -o t -- python $R/tools/gemm_diag.py $M > $R/$OUT/gemm_alias.log 2>&1); say "gemm_alias rc=$?"
  python tools/gemm_layer_report.py $(find $OUT/galias -name "*kernel_trace.csv" | head -1) $M | tee $OUT/gemm_alias_report.txt | tee -a $OUT/summary.log; rm -rf $OUT/galias ;;
pt_cycles)
  timeout 300 python tools/pt_cycles.py ${PT_CYCLES_MODES:-0,1,65,129,193,49} > $OUT/pt_cycles.txt 2>&1; say "pt_cycles rc=$?"; cat $OUT/pt_cycles.txt | tee -a $OUT/summary.log ;;
ldsprobe)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/lds_read_probe.hip -o /tmp/lds_read_probe && timeout 120 /tmp/lds_read_probe > $OUT/lds_read_probe.txt 2>&1; say "ldsprobe rc=$?"; cat $OUT/lds_read_probe.txt | tee -a $OUT/summary.log ;;
host)
  timeout 600 python tools/host_overhead.py 1000000 4000000 > $OUT/host_overhead.txt 2>&1; say "host rc=$?"; grep "^N=" $OUT/host_overhead.txt | tee -a $OUT/summary.log ;;
refatlas)
  timeout 900 python -m pytest tests/test_gpu_reference_atlas.py -m gpu -q --no-header -p no:cacheprovider -rA -s > $OUT/pytest_reference_atlas_gpu.log 2>&1; say "refatlas rc=$?"
  grep -E "reference Atlas|passed|failed|skipped" $OUT/pytest_reference_atlas_gpu.log | tee -a $OUT/summary.log ;;
gloo2)
  for extra in "" "--distinct-queries"; do
    ATLAS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --passages 2000003 --refresh-batches 0 --cpu-seconds 0 $extra > $OUT/bench_w2_gloo$extra.json 2> $OUT/bench_w2_gloo$extra.err; say "gloo2 $extra rc=$?"
    grep "^{" $OUT/bench_w2_gloo$extra.json | cut -c1-900 | tee -a $OUT/summary.log
  done ;;
libab_enc)
  timeout 900 python tools/lib_ab.py enc ${LIBAB:-r05=tools/ab/r05.so r06=tools/ab/r06.so} 5 > $OUT/lib_ab_enc.txt 2>&1; say "libab_enc rc=$?"; cat $OUT/lib_ab_enc.txt | tee -a $OUT/summary.log ;;
libab_scan)
  LIB_AB_ROWS=${LIB_AB_ROWS:-4000000} timeout 900 python tools/lib_ab.py scan ${LIBAB:-r05=tools/ab/r05.so r06=tools/ab/r06.so} 5 > $OUT/lib_ab_scan.txt 2>&1; say "libab_scan rc=$?"; cat $OUT/lib_ab_scan.txt | tee -a $OUT/summary.log ;;
enc_batch)
  timeout 900 python tools/enc_batch_size.py > $OUT/enc_batch_size.txt 2>&1; say "enc_batch rc=$?"; cat $OUT/enc_batch_size.txt | tee -a $OUT/summary.log ;;
enc_knob)
  timeout 900 python tools/enc_knob_ab.py ${ENC_KNOBS:-att0,att2,att3,noln} 3 3 > $OUT/enc_knob_ab.txt 2>&1; say "enc_knob rc=$?"; cat $OUT/enc_knob_ab.txt | tee -a $OUT/summary.log ;;
encpower)
  timeout 300 python tools/refresh_power.py random zero random zero > $OUT/refresh_power_data.txt 2>&1; say "encpower rc=$?"; cat $OUT/refresh_power_data.txt | tee -a $OUT/summary.log ;;
fullshard)
  timeout 1500 python bench.py --passages 8000000 --steps 5 --warmup 2 --cpu-seconds 0 --shard-sweep '' --batch-sweep '' --emulate-ranks '' --refresh-batches 10 --refresh-stream-seconds 5 --refresh-full-shard 4000000 > $OUT/bench_refresh_full_shard.json 2> $OUT/bench_refresh_full_shard.err; say "fullshard rc=$?"
  python - <<PY | tee -a $OUT/summary.log
import json
try:
    d = json.loads(open("$OUT/bench_refresh_full_shard.json").read().strip().splitlines()[-1])["refresh"]
    print("refresh: batch %.2f ms frac %.3f | ragged %.0f | streamed %.0f | full_shard %s" % (d["ms_per_batch"], d["roofline"]["frac"], d["ragged"]["value"], d["streamed"]["value"], json.dumps(d.get("full_shard"))))
except Exception as e:
    print("fullshard: no line", e); print(open("$OUT/bench_refresh_full_shard.err").read()[-1500:])
PY
  ;;
*) say "unknown step $STEP" ;;
esac
done
