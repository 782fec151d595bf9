import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh, gs2m_native
import diff_gaussian_rasterization as dgr
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
W, H = 1920, 1080
sc = Hh.make_scene(P, W, H, seed=0, fc=9)
g = {k: v.cuda() for k, v in sc["g"].items()}
st = Hh.settings_for(sc, "cuda")
e = torch.Tensor([])
R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
    st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
    st.projmatrix, st.tanfovx, st.tanfovy, H, W, g["shs"], 3, st.campos, False, 9)
Gc, Gb = sc["Gc"].cuda(), sc["Gb"].cuda()
def bwd():
    o = dgr._C.rasterize_gaussians_backward(st.bg, g["means3D"], radii, buffer, e, g["scales"], g["rotations"], 1.0, e,
        g["features"], st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, Gc, Gb, g["shs"], 3, st.campos, geomB, R, binB, imgB, 9)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in o]
a = bwd(); b = bwd(); c = bwd()
lay = gs2m_native.debug_layout(P, R, W, H)
al = lambda t: (-t.data_ptr()) % 256
view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
tt = view(geomB, lay.tiles_touched, P, np.uint32)
names = ["means2D", "colors", "opacities", "means3D", "cov3D", "shs", "scales", "rotations", "features"]
for x, nm in ((b, "b"), (c, "c")):
    bad = np.zeros(P, bool)
    for n_, u, v in zip(names, a, x):
        d = (u != v).reshape(P, -1).any(1)
        bad |= d
    ids = np.nonzero(bad)[0]
    print(nm, "mismatching gaussians", len(ids), "tt of them: min/mean/max", tt[ids].min() if len(ids) else 0, tt[ids].mean() if len(ids) else 0, tt[ids].max() if len(ids) else 0, "global tt mean", tt[tt>0].mean(), "max", tt.max())
    print("  ids", ids[:20].tolist(), " tt", tt[ids[:20]].tolist())
    print("  hist tt of bad:", np.bincount(np.minimum(tt[ids], 40))[:41].tolist())
print("R", R)
# locate the mismatching single-tile Gaussians inside their tile lists
pl = view(binB, lay.point_list, R, np.uint32)
tk = view(binB, lay.tile_keys, R, np.uint32)
Tn = ((W + 15) // 16) * ((H + 15) // 16)
rg = view(imgB, lay.ranges, 2 * Tn, np.uint32).reshape(Tn, 2).astype(np.int64)
nc = view(imgB, lay.n_contrib, W * H, np.uint32).reshape(H, W)
bad = np.zeros(P, bool)
for u, v in zip(a, b):
    bad |= (u != v).reshape(P, -1).any(1)
ids = np.nonzero(bad)[0]
pos_of = {}
inst = np.nonzero(np.isin(pl, ids))[0]
print("instances of bad gaussians:", len(inst))
rows_ = []
for i in inst[:4000]:
    t = int(tk[i]); p = int(i - rg[t, 0]); L = int(rg[t, 1] - rg[t, 0])
    ty, tx = divmod(t, (W + 15) // 16)
    mc = int(nc[ty*16:ty*16+16, tx*16:tx*16+16].max())
    rows_.append((int(pl[i]), t, p, L, mc))
rows_ = np.array(rows_)
print("pos%64 hist:", np.bincount(rows_[:, 2] % 64, minlength=64).tolist())
print("(maxc-1-pos) hist (distance from the last processed entry):", np.bincount(np.clip(rows_[:, 4] - 1 - rows_[:, 2], 0, 70))[:71].tolist())
print("pos>=maxc count:", int((rows_[:, 2] >= rows_[:, 4]).sum()), " tiles involved:", len(np.unique(rows_[:, 1])))
print("sample", rows_[:12].tolist())
