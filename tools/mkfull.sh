#!/bin/bash
# Build a variant library with EVERY translation unit recompiled with extra flags (a change that touches common.h's structs).
# usage: tools/mkfull.sh <name> "<extra flags>"   -> gs-2m_amd/csrc/variants/lib<name>.so
set -e
C=/root/repo/gs-2m_amd/csrc
D=$C/variants/full_$1
mkdir -p $D
objs=""
for f in api preprocess binning tile_sort radix_sort blend_fwd_q blend_bwd_q gaussian_bwd knn render_ops optim ssim texture cubemap mvs loss_ops; do
  objs="$objs $D/$f.o"
  echo "$D/$f.o: $C/$f.hip; /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -std=c++17 -Wno-inline-asm -I$C $2 -c $C/$f.hip -o $D/$f.o"
done > $D/rules.txt
python3 - "$D" <<'PY'
import sys
d = sys.argv[1]
lines = open(d + "/rules.txt").read().strip().split("\n")
with open(d + "/Makefile", "w") as f:
    f.write("all: " + " ".join(l.split(":")[0] for l in lines) + "\n")
    for l in lines:
        t, rest = l.split(": ", 1)
        dep, cmd = rest.split("; ", 1)
        f.write(f"{t}: {dep}\n\t{cmd}\n")
PY
make -s -j8 -f $D/Makefile all
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $C/variants/lib$1.so $objs
rm -rf $D
echo built $C/variants/lib$1.so
