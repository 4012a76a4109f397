#!/bin/bash
# PMC FETCH_SIZE passes only (own runs, --kernel-trace only) at 1M / 4M / 32M rows -> profiles/pmc_traffic.json (keyed by scan_kernel.h)
OUT=gpurun_out/r02p; mkdir -p $OUT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w tools/microbench.hip -o tools/libatlas_mb.so 2>&1 | tail -2
for n in 1000000 4000000 32000000; do
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$n -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $n > $GRAFT_REPO_ROOT/$OUT/pmc_$n.log 2>&1); echo "pmc $n rc=$?"
  cp $(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1) $OUT/pmc_${n}_fetch_counter_collection.csv
done
python tools/pmc_summarize.py $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
grep -h "scan_kernel" $OUT/pmc_32000000_fetch_counter_collection.csv | head -2 | cut -c1-200
rm -rf $OUT/pmc_1000000 $OUT/pmc_4000000 $OUT/pmc_32000000
