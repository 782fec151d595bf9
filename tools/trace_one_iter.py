"""ONE training iteration of a rocprofv3 kernel trace, kernel by kernel in start order with the idle gap in front of each:
python tools/trace_one_iter.py <trace dir> <anchor substring> [which iteration from the end, default 5]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
anchor, back = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]
prev_end = int(rows[a]["End_Timestamp"])
t0 = int(rows[a]["Start_Timestamp"])
tot = gap = 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"<.*", "", name)[:58]
    g = max(0, s - prev_end) / 1e3
    print("%9.1f us  +%6.1f gap  %7.1f us  %s  [grid %s]" % ((s - t0) / 1e3, g, (e - s) / 1e3, name, r.get("Grid_Size", "?")))
    tot += (e - s) / 1e3; gap += g
    prev_end = max(prev_end, e)
print("iteration: %.1f us of kernels, %.1f us of gaps, %d launches" % (tot, gap, b - a))
