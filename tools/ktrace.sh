#!/bin/bash
# rocprofv3 kernel trace of the default bench workload -> per-kernel average durations (gpurun_out/ktrace.txt)
# usage: tools/ktrace.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caller-levels --no-reference-binning "$@" > $OUT/log.txt 2>&1
python3 $R/tools/ktrace_sum.py $OUT
