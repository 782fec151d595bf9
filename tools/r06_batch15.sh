#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b15; mkdir -p $O
timeout 900 python tools/c4_host_profile.py 1200 300 > $O/c4_host_profile.txt 2>&1; grep -v amdgpu $O/c4_host_profile.txt | head -120
timeout 300 python tools/host_profile.py > $O/host_profile.txt 2>&1; grep -v amdgpu $O/host_profile.txt | head -40
