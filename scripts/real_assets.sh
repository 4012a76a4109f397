#!/bin/bash
# ONE command for the day real assets exist. Nothing here is obtainable offline: every encoder number in this repository is
# on random-init weights and synthetic token ids, and BASELINE configs[4] has never run.
#
#   ATLAS_CONTRIEVER_DIR=<facebook/contriever in the HF layout, with its tokenizer files> \
#   [ATLAS_PASSAGES=<corpus.jsonl>] [ATLAS_INDEX_DIR=<saved Atlas index> ATLAS_INDEX_SHARDS=128] [ATLAS_QUERIES=<nq-dev.jsonl>] \
#   bash scripts/real_assets.sh [out_dir]
#
# Steps (each skipped with a message when its input is missing; outputs under out_dir, default gpurun_out/real_assets):
#   1. [needs /root/reference: build container]  encoder goldens regenerated from the REAL checkpoint through the reference's own modules
#      (tests/golden/make_golden_real.py -> tests/golden/enc_real.npz; copy that file to the GPU box with the repository)
#   2. [GPU] tests/test_gpu_encoder_real.py (HIP encoder vs that fixture), tests/test_gpu_encoder_outliers.py un-skipped on the real checkpoint,
#      the retrieval-level agreement test on real weights (tests/test_gpu_encoder.py::test_retrieval_level_agreement_of_refreshed_slabs)
#   3. [GPU, ATLAS_PASSAGES] refresh passages/s INCLUDING HF tokenisation beside the TokenStore path (tools/refresh_real.py); the refreshed index is
#      saved and, when no ATLAS_INDEX_DIR is given, used by step 4
#   4. [GPU, ATLAS_QUERIES] BASELINE configs[4] without the reader: the retrieve-only loop of evaluate.py:40-83 (tools/retrieve_only.py)
#   5. [GPU] bench.py's refresh legs on the real checkpoint (bench.py reads ATLAS_CONTRIEVER_DIR itself)
# DRY RUN on stand-in assets (random weights, made-up vocabulary and corpus; proves the pipeline, nothing else):
#   python tools/make_fake_assets.py /tmp/fake && ATLAS_CONTRIEVER_DIR=/tmp/fake/contriever ATLAS_PASSAGES=/tmp/fake/passages.jsonl \
#   ATLAS_QUERIES=/tmp/fake/queries.jsonl bash scripts/real_assets.sh
set -u
OUT=${1:-gpurun_out/real_assets}; mkdir -p "$OUT"
say() { echo "$@" | tee -a "$OUT/summary.log"; }
: "${ATLAS_CONTRIEVER_DIR:?set ATLAS_CONTRIEVER_DIR to a local facebook/contriever directory}"
export ATLAS_CONTRIEVER_DIR
HAVE_GPU=$(python -c "import torch; print(int(torch.cuda.is_available()))" 2>/dev/null)
say "== real_assets: checkpoint $ATLAS_CONTRIEVER_DIR, gpu=$HAVE_GPU, passages=${ATLAS_PASSAGES:-none}, index=${ATLAS_INDEX_DIR:-none}, queries=${ATLAS_QUERIES:-none}"

if [ -d /root/reference/src ] || [ -n "${ATLAS_REFERENCE_DIR:-}" ]; then
  python tests/golden/make_golden_real.py --checkpoint "$ATLAS_CONTRIEVER_DIR" ${ATLAS_PASSAGES:+--passages "$ATLAS_PASSAGES"} > "$OUT/golden_real.log" 2>&1
  say "1. goldens from the real checkpoint through the reference modules: rc=$? $(tail -1 "$OUT/golden_real.log" | cut -c1-300)"
else
  say "1. skipped: no reference checkout on this box (run this step in the build container; tests/golden/enc_real.npz travels with the repository)"
fi
if [ "$HAVE_GPU" != "1" ]; then say "no GPU here: steps 2-5 skipped"; exit 0; fi

python -m pytest tests/test_gpu_encoder_real.py tests/test_gpu_encoder_outliers.py "tests/test_gpu_encoder.py::test_retrieval_level_agreement_of_refreshed_slabs" \
  -m gpu -q -s --no-header -rs -p no:cacheprovider > "$OUT/pytest_real.log" 2>&1
say "2. encoder tests on the real checkpoint: rc=$?"; grep -E "real checkpoint|retrieval-level agreement|passed|failed|SKIPPED" "$OUT/pytest_real.log" | cut -c1-400 | tee -a "$OUT/summary.log"

if [ -n "${ATLAS_PASSAGES:-}" ]; then
  python tools/refresh_real.py --checkpoint "$ATLAS_CONTRIEVER_DIR" --passages "$ATLAS_PASSAGES" --max-passages "${ATLAS_MAX_PASSAGES:-200000}" \
    --save-index "$OUT/index" --shards 8 > "$OUT/refresh_real.json" 2> "$OUT/refresh_real.err"
  say "3. refresh incl. HF tokenisation vs TokenStore: rc=$?"; cut -c1-1200 "$OUT/refresh_real.json" | tee -a "$OUT/summary.log"
  : "${ATLAS_INDEX_DIR:=$OUT/index}"; : "${ATLAS_INDEX_SHARDS:=8}"
else
  say "3. skipped: no ATLAS_PASSAGES"
fi
if [ -n "${ATLAS_QUERIES:-}" ] && [ -n "${ATLAS_INDEX_DIR:-}" ]; then
  python tools/retrieve_only.py --index "$ATLAS_INDEX_DIR" --shards "${ATLAS_INDEX_SHARDS:-128}" --checkpoint "$ATLAS_CONTRIEVER_DIR" --queries "$ATLAS_QUERIES" \
    --n-context 40 --batch 64 > "$OUT/retrieve_only.json" 2> "$OUT/retrieve_only.err"
  say "4. retrieve-only loop (configs[4] without the reader): rc=$?"; cut -c1-1200 "$OUT/retrieve_only.json" | tee -a "$OUT/summary.log"
else
  say "4. skipped: needs ATLAS_QUERIES and an index (ATLAS_INDEX_DIR or step 3)"
fi
python bench.py --steps 5 --warmup 2 --passages 4000000 --cpu-seconds 0 --shard-sweep '' --batch-sweep '' --emulate-ranks '' --refresh-full-shard 100000 \
  > "$OUT/bench_refresh_real.json" 2> "$OUT/bench_refresh_real.err"
say "5. bench.py refresh legs on the real checkpoint: rc=$?"; python - "$OUT/bench_refresh_real.json" <<'PY' | tee -a "$OUT/summary.log"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["refresh"]; print("   ", r["data"], "| %.0f passages/s, frac %.3f" % (r["value"], r["roofline"]["frac"]), "| full_shard", (r.get("full_shard") or {}).get("value"))
except Exception as e:
    print("    no bench line:", e)
PY
