#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06b17; mkdir -p $O
timeout 3000 python tests/ref_big.py > $O/beyond_raw.txt 2>&1; grep -E "FAIL|^ok" $O/beyond_raw.txt | cut -c1-330
python - <<'PY' > $O/gid_event.txt 2>&1
import os, sys
for p in ("/root/repo", "/root/repo/gs-2m_amd", "/root/repo/tests"):
    sys.path.insert(0, p)
import numpy as np
import helpers as Hh
from oracle import oracle
oracle.use_native_build()
os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count()))
sc = Hh.make_scene(2_000_000, 3840, 2160, seed=7, fc=9, scale_hi=0.02)
o, og = Hh.run_oracle(oracle, sc, backward=False)
gid = 1719330
print("cost", Hh.observe_event_cost(o, gid), "closest threshold event at or in front of gid", gid, ":", Hh.observe_event(o, gid, observe=False, band=1e-9))
PY
cat $O/gid_event.txt | grep -v amdgpu
