#!/bin/bash
# kernel-trace A/B of variant libraries over several bench configurations: tools/ab_configs.sh "<kernel regex>" <variant|base>...
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; shift
for cfg in "--config c3" "--config c1" "--config c2" "--config c5" "--heavy-tail 0.002:12"; do
  echo "#### $cfg"
  BENCH_ARGS="$cfg" bash $R/tools/kt_variants.sh "$K" "$@"
done
