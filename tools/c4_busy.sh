#!/bin/bash
# -> profiles/c4_kernel_time.json (see tools/c4_busy.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-5000}
OUT=/tmp/c4busy; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/c4_train_short.py $N $N > $OUT/log.txt 2>&1
grep it_per_s $OUT/log.txt
python3 $R/tools/c4_busy.py $OUT $N $R/gpurun_out/c4_kernel_time.json
