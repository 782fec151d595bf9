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
sums = Hh.run_hip_sums(sc)
for k in ("means2D", "conics", "opacities", "colors"):
    a, b = np.asarray(sums[k][gid], np.float64).reshape(-1), np.asarray(og[k][gid], np.float64).reshape(-1)
    print("sum", k, "oracle", b, "hip - oracle", a - b, "reference - oracle", np.asarray(rg[k][gid], np.float64).reshape(-1) - b)
# over all visible Gaussians: the relative error of the blend sums against the oracle's (double accumulators), HIP path and reference build
for k in ("means2D", "conics", "colors"):
    b = np.asarray(og[k], np.float64).reshape(len(og[k]), -1)
    sc_ = np.abs(b).max(1) + 1e-30
    eh = (np.abs(np.asarray(sums[k], np.float64).reshape(b.shape) - b).max(1) / sc_)[np.abs(b).max(1) > 1e-3]
    er = (np.abs(np.asarray(rg[k], np.float64).reshape(b.shape) - b).max(1) / sc_)[np.abs(b).max(1) > 1e-3]
    print(f"row-relative error of sum:{k} vs oracle: hip median {np.median(eh):.2e} p99 {np.percentile(eh, 99):.2e} p99.99 {np.percentile(eh, 99.99):.2e} | reference median {np.median(er):.2e} p99 {np.percentile(er, 99):.2e} p99.99 {np.percentile(er, 99.99):.2e}")
for k in ("scales", "rotations", "means3D"):
    print(k, "oracle", og[k][gid], "\n   hip - oracle", g[k][gid] - og[k][gid], "\n   reference - oracle", rg[k][gid] - og[k][gid], "\n   reference run 2 - reference run 1", rg2[k][gid] - rg[k][gid])
