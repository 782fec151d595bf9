"""Material-stage view at the bench workload (SURVEY.md 8(d) config C3 "+ deferred pbr.shade"): render(material_stage=True) ->
pbr_render (build_mips on a 512^2 environment light + deferred split-sum shading) -> L1 on the shaded image -> backward to
the Gaussians AND the light."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import gs2m_synth as S
from gs2m_scene import GaussianParams, PipelineParams, Camera
from gaussian_renderer import render
from pbr import CubemapLight, get_brdf_lut, pbr_render

P, W, H = 1_000_000, 1920, 1080
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
bg = torch.zeros(3, device=dev)


class Scene:
    cubemap = CubemapLight(base_res=512)
    brdf_lut = get_brdf_lut().to(dev)


scene = Scene()
rays = F.normalize(cam.get_rays().view(-1, 3), p=2, dim=-1)
gt = torch.rand(3, H, W, device=dev)


def stage(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("%-58s %.3f ms" % (name, (time.perf_counter() - t0) / n * 1e3))


def build_only():
    scene.cubemap.build_mips()


def build_fwd_bwd():
    scene.cubemap.build_mips()
    loss = sum(s.sum() for s in scene.cubemap.specular) + scene.cubemap.diffuse.sum()
    scene.cubemap.base.grad = None
    loss.backward()


def view():
    for t in pc.parameters():
        t.grad = None
    scene.cubemap.base.grad = None
    out = render(cam, pc, pipe, bg, material_stage=True)
    pkg = pbr_render(scene, cam, rays, out, metallic=False, fused="--unfused-shade" not in sys.argv)
    img = torch.where(out["normal_mask"], pkg["render_rgb"].permute(2, 0, 1), bg[:, None, None])
    ((img - gt).abs().mean()).backward()


stage("CubemapLight.build_mips, 512^2 base (6 levels), forward", build_only)
stage("... forward + backward", build_fwd_bwd)
stage("material-stage view: render + pbr_render + L1, fwd+bwd", view)
