"""Work distribution of the bench workload (GPU box): instances per Gaussian, list length per tile,
fraction of (instance, quadrant) rows that are valid.  Guides load-balancing decisions."""
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

P, W, H, fc = 1_000_000, 1920, 1080, 9
dev = "cuda"
cam = S.make_camera(W, H)
g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam, seed=0).items()}
e = torch.Tensor([])
gs2m_native.set_bwd_impl(1)  # the statistics below are those of the tile-list kernels (slot-major rows with validity bytes)
R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
    torch.zeros(3, device=dev), g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"],
    cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"], H, W, g["shs"], 3,
    cam["campos"].to(dev), False, fc)
torch.cuda.synchronize()
lay = gs2m_native.debug_layout(P, R, W, H)
view = lambda t, off, n, dt: t[off:off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
al = lambda t: (-t.data_ptr()) % 256
tt = view(geomB, al(geomB) + lay.tiles_touched, P, np.uint32).astype(np.int64)
Tn = ((W + 15) // 16) * ((H + 15) // 16)
rg = view(imgB, al(imgB) + lay.ranges, Tn * 2, np.uint32).reshape(Tn, 2).astype(np.int64)
ll = rg[:, 1] - rg[:, 0]
q = [50, 90, 99, 99.9, 99.99, 100]
print("R", R, "visible", int((radii > 0).sum().item()), "emitting", int((tt > 0).sum()))
print("instances/Gaussian percentiles", q, np.percentile(tt[tt > 0], q))
sg = view(geomB, al(geomB) + lay.sorted_gid, P, np.uint32)
tts = tt[sg]  # in depth-sorted order = gaussian_bwd's thread order
w = tts[: (P // 64) * 64].reshape(-1, 64)
print("per-wave64 (depth order): mean of max", w.max(1).mean(), "mean of mean", w.mean(1).mean(), "max of max", w.max())
print("tile list length percentiles", q, np.percentile(ll, q), "mean", ll.mean())
print("sum over tiles of ceil(len/32)*32 / R =", (np.ceil(ll / 32) * 32).sum() / R)
print("observe>0", int((observe > 0).sum().item()))

# ---- per-pixel / per-quadrant work and valid partial rows (needs a backward) ----
N = W * H
nc = view(imgB, al(imgB) + lay.n_contrib, N, np.uint32).reshape(H, W).astype(np.int64)
print("n_contrib: mean", nc.mean(), "p99", np.percentile(nc, 99), "sum (pairs traversed, pixel-exact)", nc.sum())
Hq, Wq = H // 8 * 8, W // 8 * 8
qmax = nc[:Hq, :Wq].reshape(Hq // 8, 8, Wq // 8, 8).max((1, 3))
print("sum over 8x8 quadrants of max n_contrib * 64 =", qmax.sum() * 64, " mean quadrant max", qmax.mean())
tmax = nc[: H // 16 * 16, : W // 16 * 16].reshape(H // 16, 16, W // 16, 16).max((1, 3))
print("mean tile max n_contrib", tmax.mean(), " (list mean", ll.mean(), ")")
keep = []
orig = dgr._Alloc._alloc
def spy(self, n, u):
    r = orig(self, n, u); keep.append(self.tensor); return r
dgr._Alloc._alloc = spy
Gc, Gb = S.make_upstream_grads(H, W, seed=0)
for impl in (1, 0):
    gs2m_native.set_bwd_impl(impl)
    keep.clear()
    dgr._C.rasterize_gaussians_backward(
        torch.zeros(3, device=dev), g["means3D"], radii, buffer, e, g["scales"], g["rotations"], 1.0, e, g["features"],
        cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"], Gc.to(dev), Gb.to(dev),
        g["shs"], 3, cam["campos"].to(dev), geomB, R, binB, imgB, fc)
    torch.cuda.synchronize()
    t = keep[-1]
    rpi = 4 if impl == 1 else 1
    rowf = 20 if impl == 1 else (20 if fc <= 9 else 24)
    a0 = (-t.data_ptr()) % 256
    rb = (R * rpi * rowf * 4 + 255) // 256 * 256
    v = t[a0 + rb: a0 + rb + R * rpi].cpu().numpy()
    if impl == 1:
        rws = t[a0: a0 + R * rpi * rowf * 4].view(torch.float32).view(R * rpi, rowf)
        vm = torch.from_numpy(v != 0).to(dev)
        nz = (rws != 0).any(1) & vm
        print("impl 1: valid rows that are entirely zero:", int((vm & ~nz).sum().item()), "of", int(vm.sum().item()))
    print(f"impl {impl}: valid rows {int((v != 0).sum())} of {R * rpi}; instances with any valid row",
          int((v.reshape(R, rpi) != 0).any(1).sum()))
