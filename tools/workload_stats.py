"""Work distribution of the bench workload (run on the GPU box): instances per Gaussian, list length per tile, the
per-quadrant lists of the blend kernels, their gradient rows, and the LANE EFFICIENCY of the blend kernels -- the share
of evaluated (pixel, list entry) pairs whose alpha reaches 1/255.  The pair census runs the
reference's alpha evaluation (forward.cu:326-337) in PyTorch on the device over a sample of the tiles.
usage: python tools/workload_stats.py [config c2|c3|c5] [tiles sampled]   -> text on stdout (profiles/r03_workload.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import gs2m_native
import gs2m_synth as S
import diff_gaussian_rasterization as dgr

CONFIGS = {"c2": (500_000, 1920, 1080, 5), "c3": (1_000_000, 1920, 1080, 9), "c5": (2_000_000, 1920, 1080, 9)}
P, W, H, fc = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
NSAMPLE = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = "cuda"
cam = S.make_camera(W, H)
g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam, seed=0).items()}
e = torch.Tensor([])
R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
    torch.zeros(3, device=dev), g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"],
    cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"], H, W, g["shs"], 3,
    cam["campos"].to(dev), False, fc)
torch.cuda.synchronize()
lay = gs2m_native.debug_layout(P, R, W, H)
al = lambda t: (-t.data_ptr()) % 256
view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
tt = view(geomB, lay.tiles_touched, P, np.uint32).astype(np.int64)
Tn = ((W + 15) // 16) * ((H + 15) // 16)
tiles_x = (W + 15) // 16
rg = view(imgB, lay.ranges, Tn * 2, np.uint32).reshape(Tn, 2).astype(np.int64)
ll = rg[:, 1] - rg[:, 0]
q = [50, 90, 99, 99.9, 99.99, 100]
print(f"workload {P} Gaussians {W}x{H} fc {fc}: R {R} visible {int((radii > 0).sum().item())} emitting {int((tt > 0).sum())}")
print("instances/Gaussian percentiles", q, np.percentile(tt[tt > 0], q))
w = tt[: (P // 64) * 64].reshape(-1, 64)  # index order = the emit / row-sum waves (round 5 on)
print("per-wave64 (index order): mean of max", w.max(1).mean(), "mean of mean", w.mean(1).mean(), "max of max", w.max())
print("tile list length percentiles", q, np.percentile(ll, q), "mean", ll.mean())
print("observe>0", int((observe > 0).sum().item()))
N = W * H
nc = view(imgB, lay.n_contrib, N, np.uint32).reshape(H, W).astype(np.int64)
print("n_contrib: mean", nc.mean(), "p99", np.percentile(nc, 99), "sum (pixel x instance pairs the reference traverses)", nc.sum())

# ---- per-quadrant lists (second binning level): the sorted values carry the 4-bit quadrant mask above the id ----
pl = view(binB, lay.point_list, R, np.uint32)
mask = (pl >> 28).astype(np.int64)
gid = (pl & 0x0FFFFFFF).astype(np.int64)
pop = np.array([bin(m).count("1") for m in range(16)])[mask]
print("(instance, quadrant) list entries = gradient rows:", int(pop.sum()), f"= {pop.sum() / R:.3f} per tile instance; instances with no quadrant:", int((pop == 0).sum()))
print("quadrants per instance histogram 0..4:", np.bincount(pop, minlength=5).tolist())

# ---- lane efficiency: evaluated (pixel, entry) pairs of the quadrant kernels vs pairs with alpha >= 1/255 ----
rec = view(geomB, lay.rec, P * 32, np.float32).reshape(P, 32)
rng = np.random.default_rng(0)
sample = rng.choice(np.nonzero(ll > 0)[0], size=min(NSAMPLE, int((ll > 0).sum())), replace=False)
ev = live = ev84 = 0
nent = dead_ent = 0
hist_live = np.zeros(65, np.int64)
px = torch.arange(16, device=dev, dtype=torch.float32)
for t in sample:
    lo, hi = rg[t]
    ids = torch.from_numpy(gid[lo:hi]).to(dev)
    m = torch.from_numpy(mask[lo:hi]).to(dev)
    r = torch.from_numpy(rec[gid[lo:hi], :6]).to(dev)  # x, y, A, B, C, opacity
    x0, y0 = (t % tiles_x) * 16, (t // tiles_x) * 16
    dx = r[:, 0, None, None] - (x0 + px)[None, None, :]
    dy = r[:, 1, None, None] - (y0 + px)[None, :, None]
    power = -0.5 * (r[:, 2, None, None] * dx * dx + r[:, 4, None, None] * dy * dy) - r[:, 3, None, None] * dx * dy
    a = (power <= 0) & (torch.clamp(r[:, 5, None, None] * torch.exp(power), max=0.99) >= 1.0 / 255.0)  # (n, 16, 16)
    inimg = ((y0 + px)[:, None] < H) & ((x0 + px)[None, :] < W)
    for qd in range(4):
        sel = ((m >> qd) & 1).bool()
        ys, xs = slice(8 * (qd >> 1), 8 * (qd >> 1) + 8), slice(8 * (qd & 1), 8 * (qd & 1) + 8)
        aq = a[sel][:, ys, xs] & inimg[ys, xs]
        ev += int(sel.sum()) * 64
        live += int(aq.sum())
        cnt = aq.sum((1, 2))
        nent += int(sel.sum())
        dead_ent += int((cnt == 0).sum())
        hist_live += np.bincount(cnt.cpu().numpy(), minlength=65)
        # 8x4 units: a half of the quadrant is evaluated only if the entry has a live pixel in it (lower bound on an exact test)
        ev84 += int(aq[:, :4].any((1, 2)).sum() + aq[:, 4:].any((1, 2)).sum()) * 32
print(f"lane efficiency over {len(sample)} sampled tiles: {live} live of {ev} evaluated pixel x entry pairs = {live / max(ev, 1):.3f}"
      f"; with 8x4 units at most {ev84} pairs evaluated ({ev84 / max(ev, 1):.3f} of today's)")
print(f"list entries without ANY pixel of alpha >= 1/255 in their quadrant: {dead_ent} of {nent} = {dead_ent / max(nent, 1):.3f}")
print("live pixels per (entry, quadrant), cumulative share at 0, 4, 8, 16, 32, 48, 64:", [round(float(hist_live[:k + 1].sum()) / max(nent, 1), 3) for k in (0, 4, 8, 16, 32, 48, 64)])
