#!/bin/bash
# usage: ab_stage.sh rounds "envA" "envB" ... : prints all stage times (speed experiments only; results may be wrong)
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=$1; shift
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    eval "$v python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-caller-levels" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['stages_ms']
print('%-46s %.3f ms | ' % ('''$v'''[-46:], d['ms_per_step']) + ' '.join('%s %.3f' % (k[:6], v) for k, v in s.items()))"
  done
done
