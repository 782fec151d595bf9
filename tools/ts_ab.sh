cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tile_sort_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ts1.log
python tools/ts_micro.py 8160 200 460 | grep -v amdgpu >> gpurun_out/ts1.log
python tools/ts_micro.py 8160 400 900 | grep -v amdgpu >> gpurun_out/ts1.log
python tools/ts_micro.py 3000 300 900 | grep -v amdgpu >> gpurun_out/ts1.log
python tools/ts_micro.py 1813 100 1376 | grep -v amdgpu >> gpurun_out/ts1.log
for pol in 0 1 2; do echo "c4 policy $pol" >> gpurun_out/ts1.log; GS2M_TS_POLICY=$pol python tools/c4_profile.py run bench_data/c4_geom.npz 50 2>&1 | grep "^call" >> gpurun_out/ts1.log; done
for c in c5 c3; do for pol in 0 1; do echo "$c policy $pol" >> gpurun_out/ts1.log; GS2M_TS_POLICY=$pol python bench.py --config $c --steps 50 --warmup 20 --no-cpu-baseline --no-caller-levels --no-reference-binning 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'])" >> gpurun_out/ts1.log; done; done
cat gpurun_out/ts1.log
