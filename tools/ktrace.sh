#!/bin/bash
# rocprofv3 kernel trace of the default bench workload -> per-kernel average durations (gpurun_out/ktrace.txt)
# usage: tools/ktrace.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caller-levels "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    d[row["Kernel_Name"].split("(")[0][:70]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) // 4:]  # skip warm-up launches
    print("%-72s n %4d avg %8.2f us min %8.2f" % (k, len(v), sum(v2) / len(v2), min(v2)))
PY
