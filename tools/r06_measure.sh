#!/bin/bash
# round-6 measurement bundle on the GPU box: profile passes (kernel trace + counters), the bench lines of every configuration, the
# heavy-tailed scene, the random sweep against the reference build, sizes beyond the configurations, the error tail, the C4 loop's
# kernel time.  Outputs under gpurun_out/r06/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
timeout 900 bash tools/profile.sh > $O/profile_sh.log 2>&1
python tools/summarize_prof.py r06 > $O/summarize.log 2>&1
python bench.py > $O/r06_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_form.json 2>> $O/bench.err
for c in c1 c2 c5; do python bench.py --config $c > $O/r06_bench_$c.json 2>> $O/bench.err; done
bash tools/c4_busy.sh 5000 > $O/c4_busy.txt 2>&1; cp $R/gpurun_out/c4_kernel_time.json $R/profiles/c4_kernel_time.json
for i in 1 2 3 4; do python bench.py --config c4 $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > $O/r06_bench_c4_$i.json 2>> $O/bench.err; done
python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-caller-levels --no-reference-binning --heavy-tail 0.001:30 > $O/r06_bench_heavy_tail.json 2>> $O/bench.err
( timeout 1500 python tests/ref_report.py --sweep 300 ; timeout 2400 python tests/ref_report.py --sweep 700 300 ; timeout 1500 python tests/ref_report.py --sweep 300 1000 --precomputed ; timeout 900 python tests/ref_special_sizes.py ) > $O/r06_reference_sweep_raw.txt 2>&1
timeout 1500 python tests/ref_big.py > $O/r06_beyond_configs_raw.txt 2>&1
bash tools/c4_iter_trace.sh 1600 adam_kernel > $O/r06_c4_iteration.txt 2>&1
python3 tools/trace_one_iter.py /tmp/c4it adam 5 > $O/r06_c4_one_iteration.txt 2>&1
python tools/workload_stats.py c3 400 > $O/r06_workload.txt 2>&1
grep -E "^sweep|FAIL|ok|all" $O/r06_reference_sweep_raw.txt | tail -20
grep -E "FAIL|ok" $O/r06_beyond_configs_raw.txt | tail
for f in $O/r06_bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d.get("ms_per_step"), d.get("value"), d.get("unit"), d.get("reference_binning_ms_per_step"), (d.get("roofline") or {}).get("frac"), d.get("gpu_busy_frac"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
