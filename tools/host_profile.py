import os, sys, time, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
import torch
import gs2m_synth as S
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_gaussians
P, W, H, fc = 2000, 64, 64, 9
dev = "cuda"
cam = S.make_camera(W, H)
g = S.make_gaussians(P, cam, seed=0)
Gc, Gb = S.make_upstream_grads(H, W, seed=0); Gc, Gb = Gc.to(dev), Gb.to(dev)
prm = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
means2D = torch.zeros(P, 4, device=dev, requires_grad=True)
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
    scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(dev), projmatrix=cam["projmatrix"].to(dev), sh_degree=3, campos=cam["campos"].to(dev), prefiltered=False, feature_count=fc)
empty = torch.Tensor([])
leaves = [prm["means3D"], means2D, prm["shs"], prm["opacities"], prm["scales"], prm["rotations"], prm["features"]]
def step():
    for t in leaves: t.grad = None
    color, radii, observe, buffer = rasterize_gaussians(prm["means3D"], means2D, prm["shs"], empty, prm["opacities"], prm["scales"], prm["rotations"], empty, prm["features"], st)
    torch.autograd.backward([color, buffer], [Gc, Gb])
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 500 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
