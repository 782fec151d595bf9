#!/bin/bash
# rocprofv3 kernel trace of the reference build (oracle/_ref: the reference's kernels through hipify-perl) at the bench workloads
# -> gpurun_out/ref_kernels.txt (per-kernel average time): what each stage costs in a hipify port, next to this repository's.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tests/ref_report.py --full --c3-only > $OUT/log.txt 2>&1
python3 - <<PY > $R/gpurun_out/ref_kernels.txt
import csv, glob, collections
f = sorted(glob.glob("$OUT/*/*_kernel_trace.csv"))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-92s n %4d  median %9.1f us  total %9.1f us" % (k, len(v), v[len(v) // 2], sum(v)))
PY
head -30 $R/gpurun_out/ref_kernels.txt
