#!/bin/bash
# PMC counters (two separate passes) of an arbitrary python tool, per kernel: tools/pmc_any.sh <script.py> [args...] -> gpurun_out/pmc_any.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_any
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 $R/"$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/b -- python3 $R/"$@" > $OUT/b.log 2>&1
python3 - <<PY > $R/gpurun_out/pmc_any.txt
import csv, glob, collections, re
for sub in ("a", "b"):
    fs = glob.glob("$OUT/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no counters"); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
        k = (m.group(1) + (m.group(2) or "")) if m else n[:40]
        d[k + " g" + r.get("Grid_Size", r.get("Grid_Size_X", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(d.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", [0])))):
        print("%-46s" % k[:46], "  ".join("%s %.3g" % (n.replace("SQ_", ""), sum(v) / len(v)) for n, v in sorted(c.items())))
PY
head -40 $R/gpurun_out/pmc_any.txt
