"""How many gradient elements of the full-size bench workloads lie outside 1e-3 relative (+ floor) of the reference build's?
(what the proofs in tests/test_reference_gpu.py have to cover).  python tests/exception_census.py [P] [fc]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as Hh
from oracle import reference
import gs2m_native

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fc = int(sys.argv[2]) if len(sys.argv) > 2 else 9
sc = Hh.make_scene(P, 1920, 1080, seed=0, fc=fc)
r, rg = Hh.run_oracle(reference, sc)
for mode in (False, True):
    gs2m_native.set_reference_binning(mode)
    out, g = Hh.run_hip(sc)
    sums = Hh.run_hip_sums(sc)
    gs2m_native.set_reference_binning(False)
    print("reference binning" if mode else "default binning")
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "features"):
        chain = k in ("scales", "rotations")
        frac, worst, floor = Hh.grad_stats(g[k], rg[k], 1e-3, 1e-4 if chain else 1e-5)
        a = g[k].reshape(g[k].shape[0], -1).astype(np.float64); b = rg[k].reshape(a.shape).astype(np.float64)
        rows = int(((np.abs(a - b) > 1e-3 * np.abs(b) + floor).any(1)).sum())
        print(f"  {k:10s} elements outside {frac * a.size:8.0f} ({frac:.2e})  rows {rows:6d}  max-norm {worst:.2e}")
    for k in ("means2D", "conics", "opacities", "colors", "features"):
        ref = rg[k].reshape(sums[k].shape)
        frac, worst, floor = Hh.grad_stats(sums[k], ref, 1e-3, 1e-4 if k == "conics" else 1e-5)
        a = sums[k].reshape(sums[k].shape[0], -1).astype(np.float64); b = ref.reshape(a.shape).astype(np.float64)
        rows = int(((np.abs(a - b) > 1e-3 * np.abs(b) + floor).any(1)).sum())
        print(f"  sum:{k:8s} elements outside {frac * a.size:8.0f} ({frac:.2e})  rows {rows:6d}  max-norm {worst:.2e}")
