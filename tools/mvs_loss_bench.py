"""multi_view_loss as a whole (neighbour render + geometric terms + photometric term) forward + backward at 1920x1080 on the
synthetic surface scene, fused photometric core against the op-by-op one."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import gs2m_train, gs2m_mvs
from gs2m_model import GaussianModel, OptimizationParams
from gs2m_scene import PipelineParams
from gaussian_renderer import render

W, H = 1920, 1080
scene = gs2m_train.synthetic_scene(n_true=400_000, n_views=12, W=W, H=H, init_frac=0.5)
cams, gts, pts, cols, extent = scene
m = GaussianModel(3)
m.create_from_pcd(pts, cols, extent)
class A(OptimizationParams):
    prune_init_points = False
m.training_setup(A)
mv = gs2m_mvs.MultiViewParams()
mv.multi_view_max_dist, mv.multi_view_max_angle = 8.0, 35
msc = gs2m_mvs.MultiViewScene(cams, gts, m, mv)
pipe, bg = PipelineParams(), torch.zeros(3, device="cuda")
print("points", m.get_xyz.shape[0])


def run(fused, with_loss=True):
    for p in m.parameters():
        p.grad = None
    out = render(cams[0], m, pipe, bg, True, False, sobel_normal=False)
    if with_loss:
        l = gs2m_mvs.multi_view_loss(msc, cams[0], mv, out, pipe, bg, False, render, fused=fused, rng=random.Random(1))
    else:
        l = out["depth_map"].sum() + out["normal_map"].sum()
    l.backward()


for name, args in (("render fwd+bwd alone", (True, False)), ("+ multi_view_loss, fused photometric core", (True, True)), ("+ multi_view_loss, op by op", (False, True))):
    for _ in range(3):
        run(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run(*args)
    torch.cuda.synchronize()
    print("%-46s %.3f ms" % (name, (time.perf_counter() - t0) / 10 * 1e3))
