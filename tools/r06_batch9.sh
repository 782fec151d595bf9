#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b9; mkdir -p $O
timeout 900 python tools/c4_torch_profile.py 1200 8 > $O/torch_profile.txt 2>&1; tail -100 $O/torch_profile.txt
