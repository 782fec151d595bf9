"""Fused deferred shading (pbr_shading_fused) forward + backward at 1080p on a SMOOTH G-buffer (a rendered surface: neighbouring
pixels share texels) and on a noisy one (every pixel its own texels), against the op-by-op pbr_shading."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
from pbr import CubemapLight, get_brdf_lut, pbr_shading, pbr_shading_fused

H, W = 1080, 1920
dev = "cuda"
g = torch.Generator().manual_seed(0)
yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
smooth_n = torch.nn.functional.normalize(torch.stack([xx, yy, 1.2 - xx * xx - 0.5 * yy * yy], dim=-1), dim=-1)
noisy_n = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)
v = torch.nn.functional.normalize(torch.stack([0.3 * xx, 0.3 * yy, torch.ones_like(xx)], dim=-1), dim=-1).to(dev)
smooth_r = (0.5 + 0.45 * torch.sin(3.0 * xx + 1.0) * torch.cos(2.0 * yy)).clamp(0.04, 1.0)[..., None]
noisy_r = 0.04 + 0.96 * torch.rand(H, W, 1, generator=g)
albedo = torch.rand(H, W, 3, generator=g).to(dev).requires_grad_(True)
metal = torch.rand(H, W, 1, generator=g).to(dev)
light = CubemapLight(base_res=512)
light.build_mips()
for s in light.specular:
    s.retain_grad()
lut = get_brdf_lut().to(dev)
for gname, n, r in (("smooth G-buffer", smooth_n, smooth_r), ("noisy G-buffer", noisy_n, noisy_r)):
    n, r = n.to(dev).contiguous(), r.to(dev).contiguous()
    for name, fn in (("op by op", lambda: pbr_shading(light, n, v, albedo, r, metallic=metal, occlusion=torch.ones_like(r), irradiance=torch.zeros_like(r), brdf_lut=lut)),
                     ("fused", lambda: pbr_shading_fused(light, n, v, albedo, r, metallic=metal, brdf_lut=lut))):
        def run():
            out = fn()["render_rgb"]
            torch.autograd.grad(out.sum(), [albedo, light.diffuse] + list(light.specular), retain_graph=True)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("%-16s %-9s fwd+bwd %.3f ms" % (gname, name, e0.elapsed_time(e1) / 10))
