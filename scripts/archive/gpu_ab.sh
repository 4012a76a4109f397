#!/bin/bash
# A/B of two builds of the library on the same box, un-profiled, alternating (refresh batch 512 x 128 tokens fp16)
export TMPDIR=/tmp
A=${A:-atlas_amd/lib/libatlas_hip_base.so}; B=${B:-atlas_amd/lib/libatlas_hip.so}
for r in 1 2 3; do
  for so in $A $B; do
    echo -n "$(basename $so): "; ATLAS_HIP_SO=$PWD/$so SECS=${SECS:-3} python tools/enc_sustained.py 2>&1 | grep "ms/batch" | awk '{s+=$4; n++} END {printf "%.3f ms/batch (%d windows)\n", s/n, n}'
  done
done
