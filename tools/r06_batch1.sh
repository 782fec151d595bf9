#!/bin/bash
# round 6, batch 1 (GPU box): parity of the affine-scan backward, its time against the round-5 scans, the error tail at 2M / 4K
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
bash tools/kt_variants.sh "blend_bwd|blend_fwd|gaussian_bwd" base oldscan newton > $O/kt.txt 2>&1
cat $O/kt.txt
for L in base newton oldscan; do
  if [ "$L" == "base" ]; then LP=$R/gs-2m_amd/csrc/libgs2m_raster.so; else LP=$R/gs-2m_amd/csrc/variants/lib$L.so; fi
  GS2M_LIB=$LP timeout 900 python tests/ref_big_arbitrate.py > $O/arb_$L.txt 2>&1
  echo "== $L"; grep "row-relative" $O/arb_$L.txt
done
