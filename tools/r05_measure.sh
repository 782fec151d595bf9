#!/bin/bash
# round-5 measurement bundle on the GPU box: profile passes (kernel trace + counters), the bench lines of every configuration, the
# heavy-tailed scene, and the random sweep against the reference build.  Outputs under gpurun_out/r05/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 900 bash tools/profile.sh > $O/profile_sh.log 2>&1
python bench.py > $O/r05_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-caller-levels > $O/r05_bench_driver_form.json 2>> $O/bench.err
for c in c1 c2 c5; do python bench.py --config $c > $O/r05_bench_$c.json 2>> $O/bench.err; done
python bench.py --config c4 > $O/r05_bench_c4.json 2>> $O/bench.err
python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-caller-levels --no-reference-binning --heavy-tail 0.001:30 > $O/r05_bench_heavy_tail.json 2>> $O/bench.err
( timeout 1500 python tests/ref_report.py --sweep 300 ; timeout 2400 python tests/ref_report.py --sweep 700 300 ; timeout 1500 python tests/ref_report.py --sweep 300 1000 --precomputed ; timeout 900 python tests/ref_special_sizes.py ) > $O/r05_reference_sweep_raw.txt 2>&1
# the C4 substitute's trained model: one training view by kernel (bench_data/c4_geom.npz, written by tools/c4_profile.py train), and 300
# iterations of the training loop by kernel
if [ -f $R/bench_data/c4_geom.npz ]; then
  ( python tools/c4_profile.py run bench_data/c4_geom.npz 50 2>&1 | grep -v amdgpu.ids ; bash tools/c4_kt.sh "kernel" base 2>&1 | grep -v "^call" | head -30 ) > $O/r05_c4_view.txt 2>&1
fi
bash tools/c4_iter_trace.sh 1600 adam_kernel > $O/r05_c4_iteration.txt 2>&1
grep -E "^sweep|FAIL|ok|all" $O/r05_reference_sweep_raw.txt | tail -20
for f in $O/r05_bench*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d.get("ms_per_step"), d.get("value"), d.get("unit"), d.get("reference_binning_ms_per_step"), (d.get("roofline") or {}).get("frac"))
PY
done
