#!/bin/bash
# Build a variant library with one translation unit recompiled under extra flags.
# usage: tools/mkvariant.sh <name> <file.hip> "<extra flags>"   -> gs-2m_amd/csrc/variants/lib<name>.so
set -e
C=/root/repo/gs-2m_amd/csrc
mkdir -p $C/variants
base=$(basename $2 .hip)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -std=c++17 $3 -c $C/$2 -o $C/variants/$1_$base.o
objs=""
for f in api preprocess binning radix_sort blend_fwd blend_bwd blend_bwd_mfma blend_bwd_hyb gaussian_bwd knn; do
  if [ "$f" == "$base" ]; then objs="$objs $C/variants/$1_$base.o"; else objs="$objs $C/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $C/variants/lib$1.so $objs
echo built $C/variants/lib$1.so
