#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b11; mkdir -p $O
timeout 1200 python -m pytest tests/test_raster_gpu.py tests/test_c4_gpu.py tests/test_train_gpu.py tests/test_mvs_gpu.py tests/test_dp_gpu.py tests/test_render_ops_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
for i in 1 2 3; do python bench.py --config c4 --no-cpu-baseline > $O/bench_c4_$i.json 2> $O/bench_c4_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_c4_$i.json").read().strip().splitlines()[-1])
print("c4 run $i:", d["value"], d["unit"], {k: d["config"].get(k) for k in ("points_end", "points_max", "psnr_end")})
PY
done
bash tools/kt_variants.sh "blend_bwd|blend_fwd" base expacc > $O/kt.txt 2>&1; cat $O/kt.txt
GS2M_LIB=$R/gs-2m_amd/csrc/variants/libexpacc.so timeout 900 python tests/ref_big_arbitrate.py > $O/arb_expacc.txt 2>&1; grep -v amdgpu $O/arb_expacc.txt | cut -c1-400
GS2M_LIB=$R/gs-2m_amd/csrc/variants/libexpacc.so timeout 900 python tests/error_tail.py 5 > $O/tail_expacc.txt 2>&1; grep "pixels whose\|==" $O/tail_expacc.txt
