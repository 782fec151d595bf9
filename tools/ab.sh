#!/bin/bash
# A/B on ONE box: interleaved short bench runs; prints stage times.  usage: tools/ab.sh "<args A>" "<args B>" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${3:-2}
for i in $(seq $ROUNDS); do
  for v in "$1" "$2"; do
    if [[ "$v" == --* ]]; then pre=""; post="$v"; else pre="$v"; post=""; fi
    eval "$pre python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline $post" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['stages_ms']
print('%-40s %.3f ms  fwd %.3f bwd %.3f gbwd %.3f pre %.3f' % ('''$v''', d['ms_per_step'], s['blend_fwd'], s['blend_bwd'], s['gaussian_bwd'], s['preprocess']))"
  done
done
