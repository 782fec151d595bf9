#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b2; mkdir -p $O
timeout 900 python -m pytest tests/test_tile_sort_gpu.py tests/test_raster_gpu.py tests/test_model.py tests/test_bench_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 600 python tests/fullsize_diag.py 500000 5 1 shs > $O/c2diag.txt 2>&1; cat $O/c2diag.txt | tail -30
timeout 900 python tests/error_tail.py 25 > $O/tail.txt 2>&1; cat $O/tail.txt | tail -80
timeout 600 python tools/workload_stats.py c3 400 > $O/workload.txt 2>&1; tail -4 $O/workload.txt
