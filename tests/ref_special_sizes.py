"""Test infrastructure (not collected by pytest): unusual image shapes and densities, HIP path against the reference build
(oracle/_ref) in both binning modes; a disagreement is arbitrated through the CPU oracle (double accumulators).
    python tests/ref_special_sizes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
from oracle import oracle, reference
import gs2m_native

oracle.build()
for (P, W, H, fc, hi) in ((200_000, 3840, 2160, 9, 0.02), (50_000, 17, 2000, 5, 0.05), (50_000, 2500, 33, 10, 0.05), (300_000, 640, 360, 9, 0.01), (5, 4096, 4096, 9, 0.5)):
    sc = Hh.make_scene(P, W, H, seed=P % 97, fc=fc, scale_hi=hi)
    r, rg = Hh.run_oracle(reference, sc)
    for refbin in (False, True):
        gs2m_native.set_reference_binning(refbin)
        out, g = Hh.run_hip(sc)
        gs2m_native.set_reference_binning(False)
        try:
            assert np.array_equal(out["radii"], r.radii), "radii"
            Hh.assert_observe_close(out["observe"], r)
            Hh.assert_image_close("color", out["color"], r.color, oracle_fwd=r)
            for ch in range(10):
                Hh.assert_image_close(f"buffer[{ch}]", out["buffer"][ch], r.buffer[ch], scale=max(1.0, float(np.abs(r.buffer[ch]).max())), oracle_fwd=r)
            for k in ("shs", "opacities", "features", "means2D"):
                Hh.assert_grad_close(k, g[k], rg[k])
            print("ok  ", P, W, H, fc, "reference binning" if refbin else "default binning", "num_rendered", r.num_rendered)
        except AssertionError as e:
            # a gradient outside the bound is accepted only if its Gaussian owns a pixel ON a threshold of the blend (alpha at 1/255,
            # test_T at 1e-4: two correct fp32 implementations may decide differently there), in front of it or -- the colour behind
            # a contributor enters its dL/dalpha -- anywhere in the lists of the pixels it (nearly) contributes to
            d = np.abs(g["means2D"].astype(np.float64) - rg["means2D"])
            rows = np.nonzero((d > 1e-3 * np.abs(rg["means2D"]) + 1e-5 * np.sqrt(np.mean(rg["means2D"].astype(np.float64) ** 2))).any(1))[0]
            unproven = []
            for gid in rows[:40]:
                ev = Hh.observe_event(r, int(gid), observe=False, band=3e-4)
                for idx in np.nonzero(r.vals_sorted == gid)[0]:
                    if ev <= 3e-4:
                        break
                    tile = int(r.keys_sorted[idx] >> np.uint64(32))
                    lo_, hi_ = int(r.ranges[tile, 0]), int(r.ranges[tile, 1])
                    whole = Hh.tile_walk_events(r, tile, hi_ - lo_ - 1, observe=False)
                    tx, ty = tile % r.tiles_x, tile // r.tiles_x
                    px, py = np.meshgrid(np.arange(tx * 16, tx * 16 + 16, dtype=np.float32), np.arange(ty * 16, ty * 16 + 16, dtype=np.float32))
                    A, B, C, op = r.conic_opacity[gid]
                    dx, dy = r.means2D[gid][0] - px, r.means2D[gid][1] - py
                    power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
                    near = (power <= 0) & (op * np.exp(np.minimum(power, 0)) >= 0.99 / 255.0) & (px < r.W) & (py < r.H)
                    if near.any():
                        ev = min(ev, float(whole[near].min()))
                if ev > 3e-4:
                    unproven.append((int(gid), ev))
            verdict = "ok  " if len(rows) <= max(2, 2e-5 * P) * 2 and not unproven else "FAIL"
            print(verdict, P, W, H, fc, "reference binning" if refbin else "default binning", f"{str(e)[:90]} -- {len(rows)} Gaussians outside the element-wise bound, "
                  f"{len(rows) - len(unproven)} of them with a pixel on a threshold of the blend" + (f"; UNPROVEN {unproven[:5]}" if unproven else ""))
