"""per-kernel average durations of the newest rocprofv3 kernel trace under gpurun_out/ktrace (tools/ktrace.sh)"""
import csv, glob, collections, re, os, sys
d0 = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "ktrace")
f = sorted(glob.glob(d0 + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    n = row["Kernel_Name"]
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
    k = (m.group(1) + (m.group(2) or "")) if m else n[:60]
    d[k + " g" + row.get("Grid_Size_X", "")].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) // 4:]
    if len(v) > 8:
        print("%-60s n %4d avg %8.2f us min %8.2f" % (k[:60], len(v), sum(v2) / len(v2), min(v2)))
