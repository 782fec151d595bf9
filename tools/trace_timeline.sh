#!/bin/bash
# kernel timeline of the last bench step (GPU box): start/end offsets in microseconds, to check stream overlap
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of preprocess_kernel marks the start of the last forward
idx = max(i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + 40]:
    import re
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel|fillBuffer\w*|copyBuffer\w*|\w+Functor)", n)
    n = m.group(1) if m else n[:40]
    print("%-28s start %8.1f us  dur %7.1f us" % (n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
