"""Golden vectors from the REFERENCE BUILD (oracle/_ref/libgs2m_ref.so: the reference's own cuda_rasterizer kernels through
hipify-perl, compiled for gfx950 by oracle/ref_build/Makefile) on four small scenes -> tests/golden/ref_raster_<name>.npz.

Run ONCE on the GPU box (the reference build needs a GPU):
    gpurun -- 'python tests/golden/make_ref_golden.py gpurun_out/ref_golden'   then copy the .npz files to tests/golden/
Each file holds the scene's inputs (tests/helpers.make_scene arrays, camera, upstream gradients) and the reference build's
outputs: num_rendered and the integer state (radii, tiles_touched, point_offsets, sorted 64-bit keys and values, ranges,
n_contrib, observe, clamped), the per-Gaussian forward (depths, means2D, conic + opacity, cov3D, rgb), the images (colour,
G-buffer, final_T) and the gradient tensors.  tests/test_ref_golden.py (-m "not gpu") holds oracle/gs2m_oracle.c to them in the
build container, where no GPU exists: the CPU oracle is then pinned there too, not only on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as Hh  # noqa: E402

SCENES = {
    "small_fc9": dict(P=700, W=96, H=64, seed=101, fc=9, scale_hi=0.06),
    "ragged_deg2_fc5": dict(P=1500, W=83, H=61, seed=102, fc=5, scale_hi=0.05, sh_degree=2, bg=(0.3, 0.1, 0.2)),
    "depth_ties_fc10": dict(P=900, W=80, H=48, seed=103, fc=10, scale_hi=0.08),
    "large_and_thin_fc1": dict(P=400, W=64, H=64, seed=104, fc=1, scale_lo=0.0005, scale_hi=0.6, bg=(0.2, 0.2, 0.2)),
}


def scene(name):
    kw = dict(SCENES[name])
    sc = Hh.make_scene(kw.pop("P"), kw.pop("W"), kw.pop("H"), **kw)
    if name.startswith("depth_ties"):
        m = sc["g"]["means3D"]
        m[:, 2] = torch.round(m[:, 2] * 4) / 4  # many exactly equal view depths: ties are resolved by Gaussian id
        m[:40, 2] = -1.0
    return sc


if __name__ == "__main__":
    from oracle import reference
    assert reference.available(), "needs oracle/_ref/libgs2m_ref.so and a GPU"
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_golden")
    os.makedirs(out, exist_ok=True)
    for name in SCENES:
        sc = scene(name)
        r, rg = Hh.run_oracle(reference, sc)
        d = {"in_" + k: v.numpy() for k, v in sc["g"].items()}
        cam = sc["cam"]
        d.update(W=sc["W"], H=sc["H"], fc=sc["fc"], sh_degree=sc["sh_degree"], bg=sc["bg"].numpy(), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                 viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(),
                 Gc=sc["Gc"].numpy(), Gb=sc["Gb"].numpy(), num_rendered=r.num_rendered)
        for k in ("radii", "tiles_touched", "point_offsets", "keys_sorted", "vals_sorted", "ranges", "observe", "n_contrib", "clamped", "depths",
                  "means2D", "conic_opacity", "cov3D", "rgb", "color", "buffer", "final_T"):
            d["ref_" + k] = np.asarray(getattr(r, k))
        for k, v in rg.items():
            d["refgrad_" + k] = np.asarray(v)
        path = os.path.join(out, f"ref_raster_{name}.npz")
        np.savez_compressed(path, **d)
        print(name, "num_rendered", r.num_rendered, "visible", int((r.radii > 0).sum()), os.path.getsize(path), "bytes")
