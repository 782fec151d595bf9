#!/bin/bash
# Build a variant library with one translation unit replaced (other flags or another source revision).
# usage: tools/mkvariant.sh <name> <unit.hip> "<extra flags>" [git-rev]   -> gs-2m_amd/csrc/variants/lib<name>.so
set -e
C=/root/repo/gs-2m_amd/csrc
mkdir -p $C/variants
base=$(basename $2 .hip)
src=$C/$2
if [ -n "$4" ]; then src=$C/variants/$1_src_$base.hip; git -C /root/repo show $4:gs-2m_amd/csrc/$2 > $src; fi
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -std=c++17 -I$C $3 -c $src -o $C/variants/$1_$base.o
objs=""
for f in api preprocess binning tile_sort radix_sort blend_fwd_q blend_bwd_q gaussian_bwd knn render_ops optim ssim texture cubemap mvs loss_ops; do
  if [ "$f" == "$base" ]; then objs="$objs $C/variants/$1_$base.o"; else objs="$objs $C/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $C/variants/lib$1.so $objs
echo built $C/variants/lib$1.so
