#!/bin/bash
# usage: ab4.sh rounds "envA" "envB" ...   (all stages printed; bwd impl 2 = old quadrant kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=$1; shift
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    eval "$v python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-caller-levels \$BARGS" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['stages_ms']
print('%-60s %.3f ms ' % ('''$v''', d['ms_per_step']) + ' '.join('%s %.3f' % (k[:6], v) for k, v in s.items() if v))"
  done
done
