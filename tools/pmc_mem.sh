#!/bin/bash
# per-kernel HBM traffic + SQ counters of a short bench run (separate --pmc passes): tools/pmc_mem.sh [bench args]  -> gpurun_out/pmc_mem.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-caller-levels --no-reference-binning $@"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- $CMD > $OUT/lds.log 2>&1
python3 - <<PY > $R/gpurun_out/pmc_mem.txt
import csv, glob, collections, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq", "lds"):
    fs = glob.glob("$OUT/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no counters"); continue
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
        k = (m.group(1) + (m.group(2) or "")) if m else n[:40]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(d.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0])) / max(1, len(kv[1].get("SQ_WAVE_CYCLES", [0])))):
    a = {n: sum(v) / len(v) for n, v in c.items()}
    # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB on gfx950 with FETCH under-counting by 2 (the guide's correction)
    hbm = (2 * a.get("FETCH_SIZE", 0) + a.get("WRITE_SIZE", 0)) * 1024
    print("%-34s fetch(x2) %.1f MB write %.1f MB | %s" % (k[:34], 2 * a.get("FETCH_SIZE", 0) * 1024 / 1e6, a.get("WRITE_SIZE", 0) * 1024 / 1e6,
          "  ".join("%s %.3g" % (n.replace("SQ_", ""), v) for n, v in sorted(a.items()) if n not in ("FETCH_SIZE", "WRITE_SIZE"))))
PY
cat $R/gpurun_out/pmc_mem.txt
