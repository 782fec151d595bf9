"""Where the far tail of the blend sums' error comes from: for the Gaussians with the largest error of the HIP
path's per-Gaussian blend sums against the CPU oracle (double accumulators) at 2M Gaussians / 4K, the distance of their closest pixel
to a threshold of the blend (alpha = 1/255, power = 0, test_T = 1e-4), and the same for the reference build's worst Gaussians.
usage (GPU box): python tests/error_tail.py [top N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
from oracle import oracle, reference
oracle.use_native_build()
os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count()))
TOP = int(sys.argv[1]) if len(sys.argv) > 1 else 25
sc = Hh.make_scene(2_000_000, 3840, 2160, seed=7, fc=9, scale_hi=0.02)
r, rg = Hh.run_oracle(reference, sc)
o, og = Hh.run_oracle(oracle, sc)
sums = Hh.run_hip_sums(sc)
out, _ = Hh.run_hip(sc, backward=False)
# pixels whose blend took a different decision somewhere along the list: the rendered colour against the oracle's (float noise is ~1e-6;
# a pair taken by one implementation and not by the other moves the pixel by alpha T c ~ 1e-3 T)
co = np.asarray(o.color, np.float64)
for name, c in (("hip", out["color"]), ("reference", r.color)):
    d = np.abs(np.asarray(c, np.float64) - co).max(0).reshape(-1)
    print(f"pixels whose colour differs from the oracle's by more than 1e-5 / 3e-5 / 1e-4 / 3e-4: {name} {(d > 1e-5).sum()} / {(d > 3e-5).sum()} / {(d > 1e-4).sum()} / {(d > 3e-4).sum()} of {d.size} (max {d.max():.2e})")
for k in ("conics", "colors", "means2D"):
    b = np.asarray(og[k], np.float64).reshape(len(og[k]), -1)
    scale = np.abs(b).max(1) + 1e-30
    ok = np.abs(b).max(1) > 1e-3
    eh = np.abs(np.asarray(sums[k], np.float64).reshape(b.shape) - b).max(1) / scale
    er = np.abs(np.asarray(rg[k], np.float64).reshape(b.shape) - b).max(1) / scale
    eh[~ok] = 0; er[~ok] = 0
    print(f"== sum:{k}: p99.99 hip {np.percentile(eh[ok], 99.99):.2e} reference {np.percentile(er[ok], 99.99):.2e}; Gaussians above 1e-4: hip {(eh > 1e-4).sum()} reference {(er > 1e-4).sum()} of {ok.sum()}")
    for name, e in (("hip", eh), ("reference", er)):
        top = np.argsort(-e)[:TOP]
        near = 0
        rows = []
        for gid in top:
            cost = Hh.observe_event_cost(o, int(gid))
            ev = Hh.observe_event(o, int(gid), observe=False, band=1e-6) if cost < 400_000 else float("nan")
            near += ev <= 1e-5
            rows.append((int(gid), e[gid], eh[gid], er[gid], ev))
        print(f"  top {TOP} by {name} error: {near} own a pixel within 1e-5 of a threshold (at or in front of their entry)")
        for gid, ee, a, c, ev in rows[:12]:
            print(f"    gid {gid:8d} err hip {a:.2e} reference {c:.2e} closest threshold event {ev:.2e}")
