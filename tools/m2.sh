cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/c4_kt.sh "kernel" base 2>&1 | grep -v "^call" | head -24
for c in c3 c5 c2; do python bench.py --config $c --steps 100 --warmup 30 --no-cpu-baseline --no-caller-levels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['value'], d.get('reference_binning_ms_per_step'), d['stages_ms'])"; done
python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-caller-levels --no-reference-binning --heavy-tail 0.001:30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('heavy', d['ms_per_step'], d['stages_ms'])"
