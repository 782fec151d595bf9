R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
for L in base a1 a2 a3 a4 a5; do
  if [ $L == base ]; then LP=$R/gs-2m_amd/csrc/libgs2m_raster.so; else LP=$R/gs-2m_amd/csrc/variants/lib$L.so; fi
  echo "== $L"
  GS2M_LIB=$LP python tools/adam_bench.py 2>&1 | grep gs2m_optim
  GS2M_LIB=$LP python tools/train_step_bench.py 2>&1 | tail -2
done; done
