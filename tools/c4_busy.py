"""GPU-busy fraction of the C4 training loop: the sum of the kernel durations of the run's iterations (rocprofv3 kernel trace of
tools/c4_train_short.py <iterations>: the trajectory is deterministic, so these are the kernels `bench.py --config c4` runs) over the
UNTRACED wall time of the same iterations (the bench's own it/s: tracing stretches the wall, not the kernels).
    tools/c4_busy.sh [iterations]   ->  profiles/c4_kernel_time.json  (bench.py --config c4 prints gpu_busy_frac from it)
usage of this file: python tools/c4_busy.py <trace dir> <iterations> <out.json>"""
import collections, csv, glob, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
iters, out = int(sys.argv[2]), sys.argv[3]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
assert len(idx) >= 0.9 * iters, (len(idx), iters)  # (an iteration whose view sees nothing new skips the step; a trace may drop records)
idx = idx[-iters:]  # the run's own optimizer steps (the trace may hold a warm-up in front)
# iteration k = everything behind optimizer step k - 1 up to and including step k; the first iteration starts at the run's first kernel
# behind the step in front of it (or the first kernel of the trace)
a = idx[0]
while a > 0 and "adam_kernel" not in rows[a - 1]["Kernel_Name"] and (int(rows[a]["Start_Timestamp"]) - int(rows[a - 1]["End_Timestamp"])) < 5_000_000:
    a -= 1
seg = rows[a:idx[-1] + 1]
dur = collections.defaultdict(float)
for r in seg:
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    k = m.group(1) if m and not m.group(1).startswith(("vectorized_elementwise", "elementwise", "unrolled_elementwise", "reduce")) else "framework"
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
total = sum(dur.values())
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
traced_wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
json.dump({"iterations": iters, "optimizer_steps_seen": len(idx), "kernel_ms_total": round(total, 3), "kernel_us_per_iteration": round(1e3 * total / iters, 2), "launches_per_iteration": round(len(seg) / iters, 1),
           "traced_wall_ms": round(traced_wall, 1), "source_hash": bench.kernel_source_hash(),
           "top_ms": {k: round(v, 2) for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:14]}}, open(out, "w"), indent=1)
print(open(out).read())
