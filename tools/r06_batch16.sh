#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b16; mkdir -p $O
bash tools/kt_variants.sh "blend_bwd|blend_fwd" base b_maxilp b_memcl b_iter b_nopost b_topdown f_maxilp f_nopost f_iter base > $O/kt.txt 2>&1
cat $O/kt.txt
