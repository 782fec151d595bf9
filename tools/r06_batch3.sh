#!/bin/bash
# round 6, batch 3 (GPU box): power-threshold decision + scalar scan: full parity run, kernel times of the variants, the error tail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b3; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
bash tools/kt_variants.sh "blend_bwd|blend_fwd|preprocess" base pkscan vf > $O/kt.txt 2>&1
cat $O/kt.txt
timeout 900 python tests/error_tail.py 25 > $O/tail.txt 2>&1; grep "==\|top 25" $O/tail.txt
