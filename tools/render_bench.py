"""render() (the caller of the rasterizer: activations, normals, feature packing, G-buffer post-processing) at the
bench workload: time per view forward + backward, against the raw op.  Guides the N1 row of SURVEY.md 8(f)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import gs2m_synth as S
from gs2m_scene import GaussianParams, PipelineParams, Camera
from gaussian_renderer import render

P, W, H = 1_000_000, 1920, 1080
SOBEL = "--sobel" in sys.argv
FUSED = "--unfused" not in sys.argv
dev = "cuda"
cam0 = S.make_camera(W, H)
g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam0, seed=0).items()}
albedo = torch.rand(P, 3, device=dev) * 0.8 + 0.1
rough = torch.rand(P, 1, device=dev) * 0.8 + 0.1
metal = torch.rand(P, 1, device=dev) * 0.8 + 0.1
pc = GaussianParams.from_activated(g["means3D"], g["shs"], g["scales"], g["rotations"], g["opacities"].clamp(0.01, 0.99), albedo, rough, metal)
for t in pc.parameters():
    t.requires_grad_(True)
cam = Camera(cam0, dev)
pipe = PipelineParams()
pipe.fused_render_ops = FUSED
pipe.split_sh = FUSED
bg = torch.zeros(3, device=dev)
wts = {k: torch.rand(s, device=dev) for k, s in (("render", (3, H, W)), ("depth_map", (1, H, W)), ("normal_map", (3, H, W)),
                                                 ("albedo_map", (3, H, W)), ("roughness_map", (1, H, W)), ("local_normal_map", (3, H, W)))}

def step():
    for t in pc.parameters():
        t.grad = None
    out = render(cam, pc, pipe, bg, material_stage=True, sobel_normal=SOBEL)
    loss = sum((out[k] * w).sum() for k, w in wts.items())
    if SOBEL:
        loss = loss + (out["sobel_map"] * wts["normal_map"]).sum()
    loss.backward()

for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    step()
torch.cuda.synchronize()
print(("render(sobel_normal=%s, fused=%s) fwd+bwd incl. a weighted-sum loss: " % (SOBEL, FUSED)) + "%.3f ms per view" % ((time.perf_counter() - t0) / n * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
