"""One-off check (not collected by pytest): the binning front end far beyond the bench sizes -- 20 M Gaussians (78 125 emit
workgroups, 306 super-blocks of block sums: the two-level prefix of emit_kernel), debug mode on (the histogram kernel's side sum is
cross-checked against emit_kernel's own total): emission offsets == exclusive prefix sum of tiles_touched in depth order, the sorted
list's invariants, ranges partition [0, R).      python tests/large_p_check.py [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import helpers as Hh
import gs2m_native
import diff_gaussian_rasterization as dgr

P = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
W, H, FC = 640, 360, 5
sc = Hh.make_scene(P, W, H, seed=3, fc=FC, scale_lo=0.0005, scale_hi=0.004)
g = {k: v.cuda() for k, v in sc["g"].items()}
st = Hh.settings_for(sc, "cuda")
e = torch.Tensor([])
for refbin in (False, True):
    gs2m_native.set_reference_binning(refbin)
    gs2m_native.set_debug(True)
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, H, W, g["shs"], 3, st.campos, False, FC)
    torch.cuda.synchronize()
    gs2m_native.set_debug(False)
    lay = gs2m_native.debug_layout(P, R, W, H)
    al = lambda t: (-t.data_ptr()) % 256
    view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
    tt = view(geomB, lay.tiles_touched, P, np.uint32).astype(np.int64)
    dk = view(geomB, lay.depth_key, P, np.uint32).astype(np.int64)
    assert int(tt.sum()) == R
    tk = view(binB, lay.tile_keys, R, np.uint32).astype(np.int64)
    pl = view(binB, lay.point_list, R, np.uint32) & np.uint32(0x0FFFFFFF)
    assert np.all(np.diff(tk) >= 0)
    key = (tk << 32) | dk[pl]
    assert np.all(np.diff(key) >= 0), "depth order inside every tile"
    Tn = ((W + 15) // 16) * ((H + 15) // 16)
    rg = view(imgB, lay.ranges, 2 * Tn, np.uint32).reshape(Tn, 2).astype(np.int64)
    t = rg[:, 1] > rg[:, 0]
    o = np.argsort(rg[t, 0])
    assert rg[t, 0][o][0] == 0 and rg[t, 1][o][-1] == R and np.all(rg[t, 0][o][1:] == rg[t, 1][o][:-1])
    assert np.all(tk[rg[t, 0]] == np.nonzero(t)[0])
    assert torch.isfinite(color).all() and torch.isfinite(buffer).all()
    print(f"ok P={P} refbin={refbin} R={R} visible={int((radii > 0).sum())} big={(tt >= 512).sum()}")
gs2m_native.set_reference_binning(False)
