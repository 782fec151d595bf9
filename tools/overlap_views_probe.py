"""Two independent views in flight on two HIP streams: what does the device gain when one view's memory-bound kernels run beside
the other's ALU-bound blend kernels?  (An experiment for gradient accumulation over several views per optimizer step, where the views
of one step do not depend on each other; the headline metric runs one view at a time.)  MI355X only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gs-2m_amd"))
import torch
import gs2m_native
import gs2m_synth as S
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_gaussians

P, W, H, fc = 1_000_000, 1920, 1080, 9
dev = torch.device("cuda", 0)
cam = S.make_camera(W, H)
g = S.make_gaussians(P, cam, seed=0)
Gc, Gb = S.make_upstream_grads(H, W, seed=0)
Gc, Gb = Gc.to(dev), Gb.to(dev)
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
    bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(dev), projmatrix=cam["projmatrix"].to(dev),
    sh_degree=3, campos=cam["campos"].to(dev), prefiltered=False, feature_count=fc)
empty = torch.Tensor([])
gs2m_native.set_sort_tickets(True)  # two launches of the tile sort share the device

def leaves():
    prm = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    return prm, torch.zeros(P, 4, device=dev, requires_grad=True)

sets = [leaves(), leaves()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def view(k):
    prm, m2 = sets[k]
    for t in list(prm.values()) + [m2]:
        t.grad = None
    color, radii, observe, buffer = rasterize_gaussians(prm["means3D"], m2, prm["shs"], empty, prm["opacities"], prm["scales"],
                                                        prm["rotations"], empty, prm["features"], st, None)
    torch.autograd.backward([color, buffer], [Gc, Gb])

def run(n, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i & 1 if two else 0
        with torch.cuda.stream(streams[k]):
            view(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

import gc
gc.collect(); gc.freeze()
for two in (False, True):
    run(40, two)
for rep in range(3):
    a = run(200, False)
    b = run(200, True)
    print(f"one stream {a:.4f} ms/view ({1e3 / a:.1f} views/s)   two streams alternating {b:.4f} ms/view ({1e3 / b:.1f} views/s)", flush=True)
ga, gb = sets[0][0]["means3D"].grad, sets[1][0]["means3D"].grad
print("gradients of the two sets equal:", bool(torch.equal(ga, gb)))
