#!/bin/bash
# scripts/build_variant.sh <git rev | directory holding csrc/ and include/ as the tree does> <out.so> [extra hipcc flags]
# builds the library from another version of the sources with the product flags (atlas_amd/build.py), for tools/lib_ab.py
set -e
SRC=$1; OUT=$2; shift 2
if [ -d "$SRC" ]; then D=$SRC; else
  D=$(mktemp -d /tmp/atlas_variant.XXXX); mkdir -p $D/atlas_amd $D/include
  git archive "$SRC" atlas_amd/csrc include | tar -x -C $D
fi
mkdir -p $(dirname $OUT)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w -mllvm -amdgpu-mfma-vgpr-form=1 "$@" $D/atlas_amd/csrc/atlas_hip.hip $D/atlas_amd/csrc/encoder.hip -o $OUT
echo "$OUT <- $SRC"
