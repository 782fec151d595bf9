#!/bin/bash
# kernel-trace timing of library variants (results may be wrong: knock-outs): tools/kt_variants.sh <kernel substring> <variant|base>...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; shift
for L in "$@"; do
  if [ "$L" == "base" ]; then LP=$R/gs-2m_amd/csrc/libgs2m_raster.so; else LP=$R/gs-2m_amd/csrc/variants/lib$L.so; fi
  OUT=$R/gpurun_out/ktv_$L; rm -rf $OUT; mkdir -p $OUT
  GS2M_LIB=$LP rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-caller-levels --no-reference-binning ${BENCH_ARGS:-} > $OUT/log.txt 2>&1
  echo "== $L"; python3 $R/tools/ktrace_sum.py $OUT | grep -iE "$K"
done
