"""arbitration of tests/ref_big.py's first disagreement through the CPU oracle (double accumulators): whose dL/dscale of the Gaussian is off?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
from oracle import oracle, reference
oracle.use_native_build()
os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count()))
sc = Hh.make_scene(2_000_000, 3840, 2160, seed=7, fc=9, scale_hi=0.02)
gid = 1719330
r, rg = Hh.run_oracle(reference, sc)
r2, rg2 = Hh.run_oracle(reference, sc)
o, og = Hh.run_oracle(oracle, sc)
out, g = Hh.run_hip(sc)
for k in ("scales", "rotations", "means3D"):
    print(k, "oracle", og[k][gid], "\n   hip - oracle", g[k][gid] - og[k][gid], "\n   reference - oracle", rg[k][gid] - og[k][gid], "\n   reference run 2 - reference run 1", rg2[k][gid] - rg[k][gid])
