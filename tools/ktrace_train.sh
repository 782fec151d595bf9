#!/bin/bash
# rocprofv3 kernel trace of one training iteration (tools/train_step_bench.py) -> gpurun_out/ktrain.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ktrain
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/train_step_bench.py "$@" > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 $R/tools/iter_breakdown.py $OUT ${ANCHOR:-preprocess_kernel} > $R/gpurun_out/ktrain.txt 2>&1
cat $R/gpurun_out/ktrain.txt
