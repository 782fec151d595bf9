"""Per-iteration breakdown of a rocprofv3 --kernel-trace run of a loop that launches `preprocess_kernel` once per
iteration (bench.py, tools/train_step_bench.py, tools/render_bench.py): wall time between consecutive preprocess
launches, GPU-busy time inside it, and every kernel's share (median over the last iterations).
usage: python tools/iter_breakdown.py <trace dir> [anchor kernel substring]"""
import collections, csv, glob, re, statistics, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "preprocess_kernel"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def name(r):
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel)", n)
    if m and not m.group(1).startswith(("vectorized_elementwise", "elementwise", "unrolled_elementwise", "reduce")):
        return m.group(1)
    m2 = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)<.*?(\w+Functor|\w+_kernel_cuda|\w+Op)\b", n)
    return ("torch:" + (m2.group(2) if m2 else re.sub(r"^void ", "", n)[:48]))


starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
iters = [(starts[k], starts[k + 1]) for k in range(len(starts) // 2, len(starts) - 1)]  # second half: warmed up
wall, busy, per, cnt = [], [], collections.defaultdict(list), []
for a, b in iters:
    seg = rows[a:b]
    wall.append((int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3)
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3)
    cnt.append(len(seg))
    d = collections.defaultdict(float)
    for r in seg:
        d[name(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in d.items():
        per[k].append(v)
print("%d iterations: wall %.1f us, kernels busy %.1f us (%.0f %%), %d launches per iteration" % (
    len(iters), statistics.median(wall), statistics.median(busy), 100 * statistics.median(busy) / statistics.median(wall), statistics.median(cnt)))
for k, v in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
    if len(v) >= len(iters) // 2:
        print("  %-64s %8.1f us" % (k[:64], statistics.median(v)))
