#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b5; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
bash tools/rows_ab.sh base rows96 > $O/rows_ab.txt 2>&1; cat $O/rows_ab.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; tail -c 600 $O/bench_driver_form.json
for i in 1 2 3; do python bench.py --config c4 > $O/bench_c4_$i.json 2> $O/bench_c4_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_c4_$i.json").read().strip().splitlines()[-1])
print("c4 run $i:", d["value"], d["unit"], {k: d["config"].get(k) for k in ("points_end", "points_max", "psnr_end")})
PY
done
