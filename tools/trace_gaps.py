"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace run (median per kernel pair)."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: (re.search(r"(\w+_kernel)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1)
gaps = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    gaps[(name(a), name(b))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
    if len(v) >= 8 and sorted(v)[len(v) // 2] > 0.5:
        print("%-30s -> %-30s n=%3d  median gap %6.1f us" % (k[0], k[1], len(v), sorted(v)[len(v) // 2]))
