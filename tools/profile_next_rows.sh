#!/bin/bash
# rocprofv3 kernel traces of the "next rows" workloads (run on the GPU box through gpurun); outputs in gpurun_out/prof_next/.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_next
rm -rf $OUT; mkdir -p $OUT
for t in train_step_bench pbr_bench mvs_bench adam_bench ssim_bench; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -- python3 $R/tools/$t.py > $OUT/$t.log 2>&1
  find $OUT/$t -name "*kernel_trace.csv" -delete
done
find $OUT -name "*kernel_stats.csv"
