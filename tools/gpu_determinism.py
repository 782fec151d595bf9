"""Three forward + backward runs of the same scene: every output and gradient must be bitwise equal (fixed summation
orders everywhere; `observe` uses integer atomics).  usage: python tools/gpu_determinism.py [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sc = Hh.make_scene(P, 1920, 1080, seed=0, fc=9)
outs = []
for i in range(3):
    o, g = Hh.run_hip(sc)
    outs.append((o, g))
for i in (1, 2):
    for k in outs[0][0]:
        d = (outs[0][0][k] != outs[i][0][k]).sum()
        if d: print("run", i, "fwd", k, "mismatch", int(d))
    for k in outs[0][1]:
        a, b = outs[0][1][k], outs[i][1][k]
        d = (a != b)
        if d.any():
            idx = np.argwhere(d)
            print("run", i, "grad", k, "mismatch elems", int(d.sum()), "rows", len(np.unique(idx[:, 0])), "first", idx[:3].tolist(), a[d][:3], b[d][:3])
print("done P", P)
