#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b7; mkdir -p $O
timeout 900 python -m pytest tests/test_mvs_gpu.py tests/test_c4_gpu.py tests/test_train_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
bash tools/c4_busy.sh 5000 > $O/c4_busy.txt 2>&1; tail -30 $O/c4_busy.txt
cp $R/gpurun_out/c4_kernel_time.json $R/profiles/c4_kernel_time.json 2>/dev/null
for i in 1 2; do python bench.py --config c4 --no-cpu-baseline > $O/bench_c4_$i.json 2> $O/bench_c4_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_c4_$i.json").read().strip().splitlines()[-1])
print("c4 run $i:", d["value"], d["unit"], "busy", d.get("gpu_busy_frac"), {k: d["config"].get(k) for k in ("points_end", "points_max", "psnr_end")})
PY
done
