"""Test infrastructure: shrink a scene in which one Gaussian's gradient differs between the HIP path and the reference build to the
Gaussians of its tile lists, then greedily drop entries while the difference persists; prints what is left.
    python tests/ref_minimize.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
from oracle import reference

# the scene of tests/ref_special_sizes.py that held the example this tool was written for (a radius-2 Gaussian with one pixel at
# alpha * 255 = 1.000: a threshold event); other scenes: edit or pass P W H fc scale_hi GID
P, W, H, fc, hi, GID = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (300_000, 640, 360, 9, 0.01, 205568)
sc = Hh.make_scene(P, W, H, seed=P % 97, fc=fc, scale_hi=hi)
r, _ = Hh.run_oracle(reference, sc, backward=False)
keep = []
for idx in np.nonzero(r.vals_sorted == GID)[0]:
    tile = int(r.keys_sorted[idx] >> np.uint64(32))
    keep += list(r.vals_sorted[int(r.ranges[tile, 0]):int(r.ranges[tile, 1])])
keep = np.unique(np.array(keep, dtype=np.int64))
print("Gaussians in the tile lists of", GID, ":", len(keep))


def sub(ids):
    s = dict(sc)
    s["g"] = {k: v[torch.from_numpy(ids)].clone() for k, v in sc["g"].items()}
    return s


def differs(ids):
    s = sub(ids)
    me = int(np.nonzero(ids == GID)[0][0])
    _, rg = Hh.run_oracle(reference, s)
    _, g = Hh.run_hip(s)
    a, b = g["means2D"][me].astype(np.float64), rg["means2D"][me].astype(np.float64)
    return np.abs(a - b).max() > 1e-2 * np.abs(b).max(), a, b


ok, a, b = differs(keep)
print("sub-scene differs:", ok, a, b)
if ok:
    ids = keep
    chunk = max(1, len(ids) // 2)
    while chunk >= 1:
        i = 0
        while i < len(ids):
            cand = np.concatenate([ids[:i], ids[i + chunk:]])
            if GID in cand and len(cand) and differs(cand)[0]:
                ids = cand
            else:
                i += chunk
        chunk //= 2
    ok, a, b = differs(ids)
    print("minimal set:", len(ids), ids.tolist(), a, b)
    s = sub(ids)
    np.savez(os.path.join(ROOT, "gpurun_out", "minimal_scene.npz"), ids=ids, **{k: v.numpy() for k, v in s["g"].items()})
