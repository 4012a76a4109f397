#!/bin/bash
# Build container only: a scratch copy of the three reference files tests/test_gpu_reference_atlas.py imports, under .refstage/ (git-ignored, NOT
# gpurun-ignored: it rides along with the next gpurun snapshot). The reference's sources are never committed; remove it with `rm -rf .refstage`.
set -e
SRC=${1:-/root/reference}
mkdir -p .refstage/src
for f in __init__.py atlas.py dist_utils.py slurm.py; do
  [ -f "$SRC/src/$f" ] && cp "$SRC/src/$f" .refstage/src/ || true
done
ls -la .refstage/src
