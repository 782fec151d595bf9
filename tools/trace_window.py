"""share of every kernel in the LAST part of a rocprofv3 kernel trace, per occurrence of an anchor kernel that runs once per training
iteration: python tools/trace_window.py <trace dir> <anchor substring> [iterations from the end]"""
import collections, csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
anchor, last = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 300
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) < last + 1:
    names = collections.Counter(re.sub(r"<.*", "", r["Kernel_Name"])[:70] for r in rows)
    print("anchor seen", len(idx), "times; kernels:", names.most_common(60)); sys.exit()
a, b = idx[-last - 1], idx[-1]
seg = rows[a:b]
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 / last
d, n = collections.defaultdict(float), collections.Counter()
for r in seg:
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    k = m.group(1) if m and not m.group(1).startswith(("vectorized_elementwise", "elementwise", "unrolled_elementwise", "reduce")) else "torch:" + re.sub(r"^void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
    d[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / last
    n[k] += 1
busy = sum(d.values())
print(f"last {last} iterations: wall {wall:.1f} us per iteration, kernels busy {busy:.1f} us ({100 * busy / wall:.0f} %), {len(seg) / last:.1f} launches per iteration")
for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:45]:
    print("  %-72s %8.1f us  x %.2f" % (k[:72], v, n[k] / last))
