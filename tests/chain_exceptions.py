"""Which Gaussians carry the end-to-end dL/dscale / dL/drot exceptions?  Prints, for the random sweep of
tests/test_fuzz_gpu.py and the needle scene, the conditioning of the 2-D covariance (+0.3, as the backward uses it,
CR/backward.cu:205-207) of every Gaussian with an element outside the element-wise bound, next to the population's."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
from oracle import oracle

def cond(f):
    A, B, C = (f.conic_opacity[:, k].astype(np.float64) for k in range(3))
    det = A * C - B * B
    with np.errstate(all="ignore"):
        a, b, c = C / det + 0.3, -B / det, A / det + 0.3
        rho = (a * c - b * b) / (a * c)
    return a, b, c, rho

for case in range(24):
    rng = random.Random(9000 + case)
    P = rng.choice([1, 7, 64, 300, 1500, 4000, 9000]); W, H = rng.choice([(16, 16), (33, 17), (64, 48), (130, 70), (200, 120), (97, 255)])
    fc = rng.choice([0, 1, 3, 5, 8, 9, 10]); lo = rng.choice([0.0005, 0.005, 0.02]); hi = max(rng.choice([0.03, 0.1, 0.5, 1.2]), 2 * lo)
    seed = rng.randrange(1 << 30); rng.choice([0, 1, 2, 2]); rng.choice([False, True])
    sc = Hh.make_scene(P, W, H, seed=seed, fc=fc, scale_lo=lo, scale_hi=hi, bg=(rng.random(), rng.random(), rng.random()))
    if rng.random() < 0.3:
        sc["g"]["opacities"] = torch.clamp(sc["g"]["opacities"] * 2.5, max=0.999)
    f, gr = Hh.run_oracle(oracle, sc)
    out, g = Hh.run_hip(sc)
    hs = Hh.run_hip_sums(sc)
    a, b, c, rho = cond(f)
    # empirical amplification of the chain along the actual difference of the sums
    so = np.concatenate([gr["conics"].reshape(P, 4)[:, [0, 1, 3]], gr["means2D"][:, :2]], 1).astype(np.float64)
    sh = np.concatenate([hs["conics"].reshape(P, 4)[:, [0, 1, 3]], hs["means2D"][:, :2]], 1).astype(np.float64)
    with np.errstate(all="ignore"):
        din = np.linalg.norm(sh - so, axis=1) / (np.linalg.norm(so, axis=1) + 1e-300)
    vis = f.radii > 0
    for k in ("scales", "rotations"):
        got, ref = g[k].astype(np.float64), gr[k].astype(np.float64)
        nz = ref[ref != 0]; rms = np.sqrt(np.mean(nz * nz)) if nz.size else 0.0
        for floor_frac in (1e-5, 1e-4):
            bad = np.abs(got - ref) > 1e-3 * np.abs(ref) + floor_frac * rms
            rows = np.nonzero(bad.any(1))[0]
            if len(rows):
                with np.errstate(all="ignore"):
                    dout = np.linalg.norm(got - ref, axis=1) / (np.linalg.norm(ref, axis=1) + 1e-300)
                    amp = dout / din
                print(f"   amplification of exceptions: min {np.nanmin(amp[rows]):.3g} median {np.nanmedian(amp[rows]):.3g}; din of exceptions max {din[rows].max():.2e}; population amp p99 {np.nanpercentile(amp[vis & (din > 0)], 99):.3g} median {np.nanmedian(amp[vis & (din > 0)]):.3g}")
                print(f"case {case} P={P} {W}x{H} scales=[{lo},{hi}] {k} floor {floor_frac:g}: {len(rows)} Gaussians; rho {np.sort(rho[rows])[:8]} .. max {rho[rows].max():.3g}; "
                      f"max(a,c) {np.sort(np.maximum(a, c)[rows])[-3:]}; radii {np.sort(f.radii[rows])[-3:]}; population rho median {np.median(rho[vis]):.3g} p1 {np.percentile(rho[vis], 1):.3g}; "
                      f"worst rel {np.abs(got - ref)[rows].max() / (np.abs(ref).max() + 1e-30):.2e}")
print("done")
