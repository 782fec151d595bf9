#!/bin/bash
# PC sampling of the bench workload (rocprofv3 beta feature): per-instruction sample counts of the blend kernels.
# usage: tools/pcsample.sh [method stochastic|host_trap] [unit cycles|time] [interval]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-stochastic}; U=${2:-cycles}; I=${3:-1048576}
OUT=$R/gpurun_out/pcs_$M
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I \
  --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-caller-levels --no-reference-binning > $OUT/log.txt 2>&1
echo "rc=$?" >> $OUT/log.txt
tail -5 $OUT/log.txt
find $OUT -type f | head -20
