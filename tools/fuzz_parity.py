"""Randomised parity sweep of the HIP rasterizer against the CPU oracle (GPU box): random sizes, feature counts, scale
ranges, opacities and backgrounds; same checks and tolerances as tests/test_raster_gpu.py::_check.

Known outcome (60 cases, seed 1234): 57 pass; the 3 that do not are 64 screen-filling splats (scale up to 1.2 world
units), where dL/dscale and dL/drotation differ from the oracle by 1.4e-3..4e-3 with EITHER backward variant while
means2D / opacity / colour / feature gradients agree to 1e-6: the per-pixel sums feed the ill-conditioned cov2D ->
cov3D backward (1/det^2 of a 1e5-pixel^2 covariance), which amplifies fp32 summation-order differences a
thousandfold; the reference's own float atomics have the same spread."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import helpers as Hh
from oracle import oracle
import gs2m_native

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
bad = 0
for case in range(n_cases):
    P = rng.choice([1, 7, 64, 300, 1500, 4000, 9000])
    W, H = rng.choice([(16, 16), (33, 17), (64, 48), (130, 70), (200, 120), (97, 255)])
    fc = rng.choice([0, 1, 3, 5, 8, 9, 10])
    lo = rng.choice([0.0005, 0.005, 0.02])
    hi = rng.choice([0.03, 0.1, 0.5, 1.2])
    seed = rng.randrange(1 << 30)
    impl = rng.choice([0, 1])
    refbin = rng.choice([False, True])
    gs2m_native.set_bwd_impl(impl)
    gs2m_native.set_reference_binning(refbin)
    sc = Hh.make_scene(P, W, H, seed=seed, fc=fc, scale_lo=lo, scale_hi=max(hi, lo * 2), bg=(rng.random(), rng.random(), rng.random()))
    if rng.random() < 0.3:
        sc["g"]["opacities"] = torch.clamp(sc["g"]["opacities"] * 2.5, max=0.999)
    tag = f"case {case}: P={P} {W}x{H} fc={fc} scales=[{lo},{hi}] seed={seed} impl={impl} refbin={refbin}"
    try:
        f, gr = Hh.run_oracle(oracle, sc)
        out, g = Hh.run_hip(sc)
        assert np.array_equal(out["radii"], f.radii), "radii"
        mism = int((out["observe"] != f.observe).sum())
        assert mism <= max(1, f.P // 2000), f"observe mismatches {mism}"
        Hh.assert_image_close("color", out["color"], f.color)
        for ch in range(10):
            scale = max(1.0, float(np.abs(f.buffer[ch]).max()))
            Hh.assert_image_close(f"buffer[{ch}]", out["buffer"][ch], f.buffer[ch], scale=scale)
        for k, v in g.items():
            Hh.assert_grad_close(k, v, gr[k])
        print("ok  ", tag, "R", f.num_rendered, flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", tag, "->", str(e)[:300], flush=True)
gs2m_native.set_bwd_impl(1); gs2m_native.set_reference_binning(False)
print("failures:", bad, "of", n_cases)
