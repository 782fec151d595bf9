"""Randomised parity sweep of the HIP rasterizer against the CPU oracle: random sizes (down to one Gaussian and up to
screen-filling splats), feature counts, scale ranges, opacities, backgrounds and binning mode.
Checks: radii exact; observe within the threshold allowance; images 1e-4 with every outlier explained by an
alpha = 1/255 / T = 1e-4 threshold; the backward in two halves (blend sums element-wise 1e-3; the per-Gaussian chain
equal to the oracle's chain on the same sums); end to end, every tensor element-wise 1e-3 -- for dL/dscale and dL/drot
with the proof that every Gaussian outside the bound is ill-conditioned (helpers.assert_chain_exceptions_conditioned:
a needle-like 2-D covariance, or a measured amplification of the sums' difference by the chain; for such splats the
chain's 1/det^2 amplifies last-bit differences of the sums up to a thousandfold -- the reference's float atomics have
the same spread)."""
import os
import random

import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu


N_CASES = int(os.environ.get("GS2M_FUZZ_CASES", "100"))  # round 3 ran 24 under -m gpu and 400 once by hand; the long form is the test now


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_scene(oracle_lib, case):
    assert torch.cuda.is_available()
    import gs2m_native
    rng = random.Random(9000 + case)
    P = rng.choice([1, 7, 64, 300, 1500, 4000, 9000])
    W, H = rng.choice([(16, 16), (33, 17), (64, 48), (130, 70), (200, 120), (97, 255)])
    fc = rng.choice([0, 1, 3, 5, 8, 9, 10])
    lo = rng.choice([0.0005, 0.005, 0.02])
    hi = max(rng.choice([0.03, 0.1, 0.5, 1.2]), 2 * lo)
    seed = rng.randrange(1 << 30)
    rng.choice([0, 1, 2, 2])  # (a former blend-implementation draw: kept so that the sweep's scenes stay the same)
    refbin = rng.choice([False, True])
    gs2m_native.set_reference_binning(refbin)
    sc = Hh.make_scene(P, W, H, seed=seed, fc=fc, scale_lo=lo, scale_hi=hi, bg=(rng.random(), rng.random(), rng.random()))
    if rng.random() < 0.3:
        sc["g"]["opacities"] = torch.clamp(sc["g"]["opacities"] * 2.5, max=0.999)
    tag = f"P={P} {W}x{H} fc={fc} scales=[{lo},{hi}] seed={seed} refbin={refbin}"
    f, gr = Hh.run_oracle(oracle_lib, sc)
    out, g = Hh.run_hip(sc)
    assert np.array_equal(out["radii"], f.radii), tag
    Hh.assert_observe_close(out["observe"], f)
    Hh.assert_image_close("color " + tag, out["color"], f.color, oracle_fwd=f)
    for ch in range(10):
        scale = max(1.0, float(np.abs(f.buffer[ch]).max()))
        Hh.assert_image_close(f"buffer[{ch}] " + tag, out["buffer"][ch], f.buffer[ch], scale=scale, oracle_fwd=f)
    sums = Hh.run_hip_sums(sc)
    for k, v in g.items():
        if k in ("scales", "rotations", "means3D"):  # end to end: element-wise, and every exception must be an ill-conditioned Gaussian
            # (dL/dmeans3D runs through the same 1 / det^2 chain, backward.cu:209-280: without the proof 29 of 400 scenes failed on it)
            Hh.assert_chain_exceptions_conditioned(f, g, gr, sums, names=(k,), tag=tag)
        else:
            Hh.assert_grad_close(k, v, gr[k])
    Hh.assert_two_stage(oracle_lib, f, gr, sums)
