"""Test infrastructure (not collected by pytest): sizes beyond the BASELINE configurations, HIP path against the reference build
(oracle/_ref) with the full-size rules (helpers.check_full_size: no counted-exception budget).
    python tests/ref_big.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
from oracle import reference

for (P, W, H, fc, hi, big) in ((2_000_000, 3840, 2160, 9, 0.02, 0.0), (3_000_000, 1920, 1080, 5, 0.02, 0.0), (400_000, 3840, 2160, 9, 0.02, 0.002)):
    sc = Hh.make_scene(P, W, H, seed=7, fc=fc, scale_hi=hi)
    if big > 0:  # a heavy tail: splats over hundreds to thousands of tiles
        sel = torch.rand(P, generator=torch.Generator().manual_seed(3)) < big
        sc["g"]["scales"] = torch.where(sel[:, None], sc["g"]["scales"] * 40.0, sc["g"]["scales"])
    Hh.PROOF_BUDGET = 60_000 * 40  # 4K frames, splats over hundreds of tiles: a proof walks every instance's tile list (minutes, not the tests' seconds)
    for refbin in (False, True):
        t0 = time.time()
        try:
            r = Hh.check_full_size(reference, sc, refbin, tag=f"{P} {W}x{H} fc {fc}")
            print("ok  ", P, W, H, fc, "big" if big else "", "reference binning" if refbin else "default binning", "num_rendered", r.num_rendered, f"({time.time() - t0:.0f} s)", flush=True)
        except AssertionError as e:
            print("FAIL", P, W, H, fc, "reference binning" if refbin else "default binning", str(e)[:300], flush=True)
