#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b14; mkdir -p $O
timeout 1200 python -m pytest tests/test_mvs_gpu.py tests/test_c4_gpu.py tests/test_train_gpu.py tests/test_losses_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
for i in 1 2 3 4; do
  for mode in dev torch; do
    if [ $mode == torch ]; then export GS2M_SUBSET_TORCH=1; else unset GS2M_SUBSET_TORCH; fi
    python bench.py --config c4 --no-cpu-baseline > $O/bench_c4_${mode}_$i.json 2> $O/bench_c4_${mode}_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_c4_${mode}_$i.json").read().strip().splitlines()[-1])
print("c4 subset-$mode run $i:", d["value"], d["unit"], {k: d["config"].get(k) for k in ("points_end", "points_max", "psnr_end")})
PY
  done
done
unset GS2M_SUBSET_TORCH
bash tools/c4_busy.sh 5000 > $O/c4_busy.txt 2>&1; tail -8 $O/c4_busy.txt | head -7
