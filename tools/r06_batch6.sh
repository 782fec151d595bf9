#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b6; mkdir -p $O
./tools/micro/mfma_shapes > $O/micro.txt 2>&1; cat $O/micro.txt
bash tools/c4_iter_trace.sh 1500 > $O/c4_trace.txt 2>&1; head -60 $O/c4_trace.txt
python3 tools/trace_one_iter.py /tmp/c4it adam 5 > $O/c4_one_iter.txt 2>&1; tail -5 $O/c4_one_iter.txt
