"""Error profile of the HIP rasterizer against the CPU oracle on the parity-test scenes (GPU box): per gradient
tensor, how the element-wise error |a-b| relates to |b| and to the tensor's scale.  Used to set (and to justify)
the bounds in tests/helpers.py: assert_grad_close."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import gs2m_native
import helpers as Hh
from oracle import oracle

scenes = {
    "fc9_200x120": dict(P=4000, W=200, H=120, seed=9, fc=9, scale_hi=0.05, bg=(0.3, 0.1, 0.2)),
    "16x16": dict(P=3000, W=16, H=16, seed=16, fc=9, scale_hi=0.08, bg=(0.0, 0.5, 1.0)),
    "640x360": dict(P=3000, W=640, H=360, seed=640, fc=9, scale_hi=0.08, bg=(0.0, 0.5, 1.0)),
    "thin_large": dict(P=1500, W=160, H=128, seed=5, fc=10, scale_lo=0.0005, scale_hi=0.6, bg=(0.2, 0.2, 0.2)),
    "c1": dict(P=10000, W=256, H=256, seed=1, fc=10),
}
for name, kw in scenes.items():
    sc = Hh.make_scene(**kw)
    f, gr = Hh.run_oracle(oracle, sc)
    for impl in ("hip",):
        out, g = Hh.run_hip(sc)
        print(f"== {name} impl {impl}: color max {np.abs(out['color'] - f.color).max():.2e}  buffer max {np.abs(out['buffer'] - f.buffer).max():.2e}"
              f"  observe mism {(out['observe'] != f.observe).sum()}")
        for k, v in g.items():
            b = gr[k].astype(np.float64).ravel(); a = v.astype(np.float64).ravel()
            nz = b[b != 0]
            rms = np.sqrt((nz * nz).mean()) if nz.size else 0.0
            d = np.abs(a - b)
            line = f"  {k:10s} n {a.size:8d} rms {rms:.2e} max|b| {np.abs(b).max():.2e} maxnorm {d.max() / (np.abs(b).max() + 1e-30):.1e}"
            for ff in (1e-6, 1e-5, 1e-4, 1e-3):
                fail = d > 1e-3 * np.abs(b) + ff * rms
                line += f" | floor {ff:g}: {fail.mean():.1e}"
            # errors of the failing elements relative to the Gaussian's own gradient row (max over the row)
            rows = gr[k].reshape(gr[k].shape[0], -1).astype(np.float64)
            drow = np.abs(v.reshape(rows.shape).astype(np.float64) - rows)
            rowmax = np.abs(rows).max(1, keepdims=True)
            failrow = drow > 1e-3 * rowmax + 1e-6 * rms
            line += f" | per-row(1e-3 * row max + 1e-6 rms): {failrow.any(1).mean():.1e}"
            print(line)

        hip = Hh.run_hip_sums(sc)
        chain = oracle.backward_pergaussian(f, hip["means2D"], hip["conics"], hip["colors"])
        for k in ("conics", "colors"):
            fr, worst, fl = Hh.grad_stats(hip[k], gr[k].reshape(hip[k].shape), 1e-3, 1e-5)
            print(f"  sum:{k:8s} frac {fr:.1e} maxnorm {worst:.1e}")
        for k in ("means3D", "shs", "scales", "rotations"):
            line = f"  chain:{k:10s}"
            for rel, ff in ((1e-3, 1e-5), (1e-4, 1e-5), (1e-5, 1e-5), (1e-5, 1e-6)):
                fr, worst, fl = Hh.grad_stats(hip[k], chain[k].reshape(hip[k].shape), rel, ff)
                line += f" | rel {rel:g} floor {ff:g}: {fr:.1e}"
            print(line + f" maxnorm {worst:.1e}")
