#!/bin/bash
# A/B of gradient-row layouts: kernel times come from tools/kt_variants.sh; this adds FETCH_SIZE / WRITE_SIZE per
# launch of the two kernels that touch the rows, one --pmc pass each (MI355X_MICROARCH.md).  usage: tools/rows_ab.sh base rows96 ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for L in "$@"; do
  if [ "$L" == "base" ]; then LP=$R/gs-2m_amd/csrc/libgs2m_raster.so; else LP=$R/gs-2m_amd/csrc/variants/lib$L.so; fi
  OUT=$R/gpurun_out/rows_ab_$L; rm -rf $OUT; mkdir -p $OUT
  CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-caller-levels --no-reference-binning"
  GS2M_LIB=$LP rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
  GS2M_LIB=$LP rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
  python3 - <<PY
import csv, glob, collections, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob("$OUT/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(blend_bwd_q_kernel|gaussian_bwd_kernel)", r["Kernel_Name"])
            if m:
                d[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== $L")
for k, c in sorted(d.items()):
    a = {n: sum(v) / len(v) for n, v in c.items()}
    print("%-22s FETCH_SIZE %.0f KiB (x2 = %.1f MB)  WRITE_SIZE %.0f KiB (= %.1f MB)" % (k, a.get("FETCH_SIZE", 0), 2 * a.get("FETCH_SIZE", 0) * 1024 / 1e6, a.get("WRITE_SIZE", 0), a.get("WRITE_SIZE", 0) * 1024 / 1e6))
PY
done
