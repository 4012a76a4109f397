#!/bin/bash
# round 3, session z: the whole -m gpu suite and smoke() on the final code
OUT=gpurun_out/r03z; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee $OUT/summary.log
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rs 2>&1 | grep -E "passed|failed|error|SKIPPED" | tail -12 | tee -a $OUT/summary.log
