#!/bin/bash
# usage: ab3.sh rounds "envA" "envB" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=$1; shift
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    eval "$v python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['stages_ms']
print('%-50s %.3f ms  rng %.3f fwd %.3f bwd %.3f gbwd %.3f' % ('''$v''', d['ms_per_step'], s['ranges'], s['blend_fwd'], s['blend_bwd'], s['gaussian_bwd']))"
  done
done
