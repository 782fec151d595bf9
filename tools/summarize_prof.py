"""Condense gpurun_out/prof (tools/profile.sh) into profiles/<tag>_*.{md,json}."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    if m:
        return m.group(1) + ("<%s>" % re.search(r"ILi(\d+)E", name).group(1) if re.search(r"ILi(\d+)E", name) else "")
    return re.sub(r"\(.*", "", name)[:70]


def find(sub, pat):
    fs = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None  # gpurun merges into the directory: earlier runs' files stay


lines = [f"# rocprofv3 summary ({tag}) -- `python bench.py` default workload (1M Gaussians, 1080p, fc=9)", ""]
stats = find("trace", "*kernel_stats.csv")
if stats:
    lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats)):
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
    lines.append("")

pmc = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_sq", "pmc_lds", "pmc_mfma", "pmc_fetch", "pmc_write"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if pmc:
    names = sorted({c for k in pmc.values() for c in k})
    lines += ["## PMC counters, average per launch (separate `--pmc` passes)", "", "| kernel | " + " | ".join(names) + " |", "|---|" + "---|" * len(names)]
    avg = {}
    for k, cs in sorted(pmc.items()):
        avg[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        lines.append(f"| {k} | " + " | ".join(f"{avg[k].get(c, float('nan')):.4g}" for c in names) + " |")
    lines += ["", "HBM traffic per launch = 2 x FETCH_SIZE (gfx950 reports half the bytes of wide streaming reads, "
              "MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KiB."]
    # profiles/pmc_counters.json: what bench.py's roofline.traffic / roofline.issue are read from.  Stamped with the
    # hash of the kernel sources and the workload the passes ran on: bench.py refuses the file when either differs.
    sys.path.insert(0, ROOT)
    import bench
    workload = sys.argv[2] if len(sys.argv) > 2 else "1000000x1920x1080x9"
    kernels = {}
    for k, a in avg.items():
        role = "blend_fwd" if k.startswith("blend_fwd") else "blend_bwd" if k.startswith("blend_bwd") else k
        e = {c: a[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                               "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                               "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if c in a}
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            e.update(fetch_kib=a["FETCH_SIZE"], write_kib=a["WRITE_SIZE"], hbm_bytes=(2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024)
        kernels[role] = e
    json.dump({"source_hash": bench.kernel_source_hash(), "workload": workload, "tag": tag, "kernels": kernels},
              open(os.path.join(dst, "pmc_counters.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
