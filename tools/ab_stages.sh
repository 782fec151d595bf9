#!/bin/bash
# same-box comparison of libraries, all stage times.  usage: tools/ab_stages.sh <rounds> <lib.so>...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    GS2M_LIB=$R/$L python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']
print('%-44s %.3f ms | ' % ('$L'[-44:], d['ms_per_step']) + ' '.join('%s %.3f' % (k[:6], v) for k, v in s.items()))"
  done
done
