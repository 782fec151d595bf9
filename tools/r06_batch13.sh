#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b13; mkdir -p $O
timeout 1200 python -m pytest tests/test_render_ops_gpu.py tests/test_raster_gpu.py tests/test_mvs_gpu.py tests/test_c4_gpu.py tests/test_train_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
for i in 1 2 3; do
  for mode in noshade shade; do
    if [ $mode == shade ]; then export GS2M_SHADE_NEIGHBOUR=1; else unset GS2M_SHADE_NEIGHBOUR; fi
    python bench.py --config c4 --no-cpu-baseline > $O/bench_c4_${mode}_$i.json 2> $O/bench_c4_${mode}_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_c4_${mode}_$i.json").read().strip().splitlines()[-1])
print("c4 $mode run $i:", d["value"], d["unit"], {k: d["config"].get(k) for k in ("points_end", "points_max", "psnr_end")})
PY
  done
done
