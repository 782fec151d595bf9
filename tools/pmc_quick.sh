#!/bin/bash
# one SQ counter pass over a short bench run (GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("sq","lds"):
    f = glob.glob("$OUT/"+sub+"/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "blend" in n or "gaussian_bwd" in n:
            import re
            key = re.search(r"(\w+_kernel)", n).group(1)
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: "%.3g" % (sum(v)/len(v)) for c, v in cs.items()})
PY
