#!/bin/bash
# kernel-trace timing on the trained C4 model's geometry (bench_data/c4_geom.npz, tools/c4_profile.py): tools/c4_kt.sh <kernel regex> <variant|base>...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; shift
for L in "$@"; do
  if [ "$L" == "base" ]; then LP=$R/gs-2m_amd/csrc/libgs2m_raster.so; else LP=$R/gs-2m_amd/csrc/variants/lib$L.so; fi
  OUT=$R/gpurun_out/c4kt_$L; rm -rf $OUT; mkdir -p $OUT
  GS2M_LIB=$LP rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/c4_profile.py run $R/bench_data/c4_geom.npz 20 > $OUT/log.txt 2>&1
  echo "== $L"; grep "^call" $OUT/log.txt | cut -c1-60; python3 $R/tools/ktrace_sum.py $OUT | grep -E "$K"
  rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
done
