#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b4; mkdir -p $O
bash tools/kt_variants.sh "blend_bwd|blend_fwd|gaussian_bwd" base fwdold fwd2 fwd2w8 rows96 > $O/kt.txt 2>&1
cat $O/kt.txt
timeout 900 python tests/error_tail.py 12 > $O/tail.txt 2>&1; grep "pixels whose\|==" $O/tail.txt
