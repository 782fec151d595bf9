#!/bin/bash
# same-box comparison of library builds (stage times only).  usage: tools/abn.sh <rounds> <lib.so | base>... ; extra bench args in BENCH_ARGS
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    if [ "$L" == "base" ]; then LP=$R/gs-2m_amd/csrc/libgs2m_raster.so; else LP=$R/gs-2m_amd/csrc/variants/lib$L.so; fi
    GS2M_LIB=$LP python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-caller-levels --no-reference-binning ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']
print('%-14s %.3f ms | ' % ('$L'[-14:], d['ms_per_step']) + ' '.join('%s %.3f' % (k[:6], v) for k, v in s.items() if v))"
  done
done
