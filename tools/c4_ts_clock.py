"""tile sort phase clocks (GS2M_TS_CLOCK build) on the trained C4 model's geometry: which tiles take the tie path, where a tile's time goes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import c4_profile as C, helpers as Hh, gs2m_native
import diff_gaussian_rasterization as dgr
calls = C.calls_from_geometry(os.path.join(ROOT, "bench_data", "c4_geom.npz"))
dev = "cuda"
sc = calls[0]
g = {k: v.to(dev) for k, v in sc["g"].items()}
P, W, H, fc = g["means3D"].shape[0], sc["W"], sc["H"], sc["fc"]
st = Hh.settings_for(sc, dev)
e = torch.Tensor([])
for _ in range(3):
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, H, W, g["shs"], sc["sh_degree"], st.campos, False, fc)
torch.cuda.synchronize()
lay = gs2m_native.debug_layout(P, R, W, H)
al = lambda t: (-t.data_ptr()) % 256
up = lambda x: (x + 255) // 256 * 256
off_valA = lay.tile_keys - up(R * 4)
Tn = ((W + 15) // 16) * ((H + 15) // 16)
c = binB[al(binB) + off_valA: al(binB) + off_valA + 16 * Tn * 4].cpu().numpy().view(np.uint32).reshape(Tn, 16)
pl = binB[al(binB) + lay.point_list: al(binB) + lay.point_list + R * 4].cpu().numpy().view(np.uint32)
rg = imgB[al(imgB) + lay.ranges: al(imgB) + lay.ranges + Tn * 8].cpu().numpy().view(np.uint32).reshape(Tn, 2)
dk = geomB[al(geomB) + lay.depth_key: al(geomB) + lay.depth_key + P * 4].cpu().numpy().view(np.uint32)
ties = 0; tied_tiles = 0
for t in range(Tn):
    d = dk[pl[rg[t, 0]:rg[t, 1]] & 0x0FFFFFFF]
    k = int((d[1:] == d[:-1]).sum())
    ties += k; tied_tiles += k > 0
print(f"R {R}: tiles with equal neighbouring depths in their sorted list: {tied_tiles} of {Tn} ({ties} equal pairs)")
sel = c[:, 14] > 0
cc = c[sel]
print("phase ticks (10 ns) medians [staged, sorted, ties, payload, lists]:", np.median(cc[:, :5], axis=0).tolist(), "max", cc[:, :5].max(axis=0).tolist())
tie_cost = cc[:, 2].astype(np.int64) - cc[:, 1]
print("tiles whose tie phase took > 2 us:", int((tie_cost > 200).sum()), "median tie phase of those:", float(np.median(tie_cost[tie_cost > 200])) if (tie_cost > 200).any() else 0)
