#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b12; mkdir -p $O
for i in 1 2 3; do
  for mode in skip keep; do
    if [ $mode == keep ]; then export GS2M_KEEP_DEAD_SH=1; else unset GS2M_KEEP_DEAD_SH; fi
    python bench.py --config c4 --no-cpu-baseline > $O/bench_c4_${mode}_$i.json 2> $O/bench_c4_${mode}_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_c4_${mode}_$i.json").read().strip().splitlines()[-1])
print("c4 $mode run $i:", d["value"], d["unit"], {k: d["config"].get(k) for k in ("points_end", "points_max", "psnr_end")})
PY
  done
done
unset GS2M_KEEP_DEAD_SH
timeout 600 python tools/c4_torch_profile.py 1200 8 > $O/torch_profile.txt 2>&1; grep -E "iterations from|aten::add |_RasterizeGaussiansBackward" $O/torch_profile.txt | head -12
