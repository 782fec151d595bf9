"""Stage-by-stage comparison of the HIP pipeline with the oracle on one scene (GPU box)."""
import argparse
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
import diff_gaussian_rasterization as dgr


def view(t, off, n, dtype):
    return t[off:off + n * np.dtype(dtype).itemsize].cpu().numpy().view(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=2000)
    ap.add_argument("--W", type=int, default=128)
    ap.add_argument("--H", type=int, default=96)
    ap.add_argument("--fc", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scale_hi", type=float, default=0.05)
    a = ap.parse_args()
    print(gs2m_native.lib().gs2m_version())
    gs2m_native.set_reference_binning(True)  # compare the integer artefacts in reference mode
    sc = Hh.make_scene(a.P, a.W, a.H, seed=a.seed, fc=a.fc, scale_hi=a.scale_hi, bg=(0.1, 0.2, 0.3))
    f, gr = Hh.run_oracle(oracle, sc)
    dev = "cuda"
    g = {k: v.to(dev) for k, v in sc["g"].items()}
    st = Hh.settings_for(sc, dev)
    e = torch.Tensor([])
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, a.H, a.W, g["shs"], 3, st.campos, False, a.fc)
    torch.cuda.synchronize()
    print("R hip", R, "oracle", f.num_rendered)
    lay = gs2m_native.debug_layout(a.P, R, a.W, a.H)
    al = lambda t: (-t.data_ptr()) % 256
    P = a.P
    go, bo, io = al(geomB), al(binB), al(imgB)
    rec = view(geomB, go + lay.rec, P * 32, np.float32).reshape(P, 32)
    tt = view(geomB, go + lay.tiles_touched, P, np.uint32)
    dk = view(geomB, go + lay.depth_key, P, np.uint32)
    vis = f.radii > 0
    print("radii equal:", np.array_equal(radii.cpu().numpy(), f.radii), " tiles_touched equal:", np.array_equal(tt, f.tiles_touched))
    print("depth key equal (visible):", np.array_equal(dk[vis], f.depths[vis].view(np.uint32)), " culled keys all FFFFFFFF:", bool((dk[~vis] == 0xFFFFFFFF).all()))
    print("means2D bit-equal:", np.array_equal(rec[vis, 0:2], f.means2D[vis]))
    con = np.stack([rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5]], 1)
    print("conic/opacity max rel diff:", Hh.rel_err(con[vis], f.conic_opacity[vis]), " bit-equal:", np.array_equal(con[vis], f.conic_opacity[vis]))
    print("rgb max abs diff:", np.abs(rec[vis, 12:15] - f.rgb[vis]).max() if vis.any() else 0)
    if R == f.num_rendered and R > 0:
        pl = view(binB, bo + lay.point_list, R, np.uint32)
        print("point_list equal:", np.array_equal(pl, f.vals_sorted))
        Tn = f.tiles_x * f.tiles_y
        rg = view(imgB, io + lay.ranges, Tn * 2, np.uint32).reshape(Tn, 2)
        print("ranges equal:", np.array_equal(rg, f.ranges))
    N = a.W * a.H
    fT = view(imgB, io + lay.final_T, N, np.float32).reshape(a.H, a.W)
    nc = view(imgB, io + lay.n_contrib, N, np.uint32).reshape(a.H, a.W)
    print("final_T max abs diff:", np.abs(fT - f.final_T).max(), " n_contrib mismatches:", int((nc != f.n_contrib).sum()), "/", N)
    print("color max abs diff:", np.abs(color.cpu().numpy() - f.color).max())
    print("buffer max abs diff per channel:", np.abs(buffer.cpu().numpy() - f.buffer).reshape(10, -1).max(1))
    print("observe mismatches:", int((observe.cpu().numpy() != f.observe).sum()), "/", P, " sum hip", int(observe.sum()), "oracle", int(f.observe.sum()))
    out, grads = Hh.run_hip(sc)
    for k in grads:
        ok = {"colors": "colors", "cov3D": "cov3D"}.get(k, k)
        print(f"grad {k:10s} rel err {Hh.rel_err(grads[k], gr[ok]):.3e}   max|ref| {np.abs(gr[ok]).max():.3e}")


if __name__ == "__main__":
    main()
