#!/bin/bash
# what the driver runs at round end, on the final tree: the GPU tests, smoke(), the bench in its short form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06final; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
