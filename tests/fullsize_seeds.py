"""One-off parity run beyond the committed full-size tests: the bench workload's shape (1M Gaussians, 1080p, feature_count 9)
with OTHER seeds and from other cameras of the 8-position ring, the HIP path against the CPU oracle with the very checks of
tests/test_configs_gpu.py (_against_oracle: radii exact, observe / images with threshold-event proofs, gradients
element-wise with the conditioning proofs, the backward in its two halves).  Prints one line per case."""
import math, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import gs2m_synth as S
import helpers as Hh
import test_configs_gpu as T
from oracle import oracle as O
O.build()
W, H, P = 1920, 1080, 1_000_000
cases = [("seed %d" % s, s, 0) for s in (1, 2, 3)] + [("seed 0, ring position %d" % k, 0, k) for k in (2, 5)]
for name, seed, ring in cases:
    cam = None
    if ring:
        th = 2.0 * math.pi * ring / 8.0
        cam = S.look_at_camera(W, H, (6.0 * math.sin(th), 0.0, 6.0 - 6.0 * math.cos(th)), (0.0, 0.0, 6.0))
    sc = Hh.make_scene(P, W, H, seed=seed, fc=9, cam=cam) if cam is None else Hh.make_scene(P, W, H, seed=seed, fc=9, cam=cam)
    if cam is not None:  # the cloud of the identity-pose frustum, seen from the ring (as bench.py --ring-position does)
        sc["g"] = S.make_gaussians(P, S.make_camera(W, H), seed=seed)
    t0 = time.time()
    try:
        T._against_oracle(O, name, sc, False)
        print("%-28s PASS (%.0f s)" % (name, time.time() - t0), flush=True)
    except AssertionError as e:
        print("%-28s FAIL: %s" % (name, str(e)[:300]), flush=True)
