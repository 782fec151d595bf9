#!/bin/bash
# kernel trace of a piece of the C4 training run -> per-iteration breakdown (tools/iter_breakdown.py): tools/c4_iter_trace.sh [iterations]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/c4it; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/c4_train_short.py ${1:-1500} 5000 > $OUT/log.txt 2>&1
grep it_per_s $OUT/log.txt
python3 $R/tools/trace_window.py $OUT ${2:-adam} 300
