cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tile_sort_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/ts1.log
python tools/c4_profile.py train /tmp/c4_call.pt 5000 2>&1 | grep -v amdgpu.ids >> gpurun_out/ts1.log
for pol in -1 0 512 256; do echo "policy $pol" >> gpurun_out/ts1.log; GS2M_TS_POLICY=$pol python tools/c4_profile.py run /tmp/c4_call.pt 50 2>&1 | grep "^call" >> gpurun_out/ts1.log; done
for pol in -1 0 256 128; do echo "c5 policy $pol" >> gpurun_out/ts1.log; GS2M_TS_POLICY=$pol python bench.py --config c5 --steps 50 --warmup 20 --no-cpu-baseline --no-caller-levels --no-reference-binning 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'])" >> gpurun_out/ts1.log; done
for pol in -1 0 256; do echo "c3 policy $pol" >> gpurun_out/ts1.log; GS2M_TS_POLICY=$pol python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-caller-levels --no-reference-binning 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms'])" >> gpurun_out/ts1.log; done
cat gpurun_out/ts1.log
