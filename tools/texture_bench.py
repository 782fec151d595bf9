"""The three PBR-stage lookups at 1080p (pbr/shade.py:150-190 shapes): diffuse cube 6x16x16x3 'linear', BRDF LUT 256x256x2
'clamp', specular cube 6x512x512x3 + 5 mips 'linear-mipmap-linear' by roughness; forward and texture-gradient backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import nvdiffrast.torch as dr

H, W = 1080, 1920
dev = "cuda"
g = torch.Generator().manual_seed(0)
# smooth normals / reflection directions, as a rendered surface gives
yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
n = torch.nn.functional.normalize(torch.stack([xx, yy, 1.2 - xx * xx - 0.5 * yy * yy], dim=-1) + 0.02 * torch.randn(H, W, 3, generator=g), dim=-1)[None].to(dev).contiguous()
rough = (0.5 + 0.45 * torch.sin(3.0 * xx + 1.0) * torch.cos(2.0 * yy) + 0.03 * torch.randn(H, W, generator=g)).clamp(0.04, 1.0)[None].to(dev)  # smooth roughness map + noise
diffuse = torch.rand(1, 6, 16, 16, 3, device=dev, requires_grad=True)
spec = [torch.rand(1, 6, w, w, 3, device=dev, requires_grad=True) for w in (512, 256, 128, 64, 32, 16)]
lut = torch.rand(1, 256, 256, 2, device=dev)
uv = torch.rand(1, H, W, 2, device=dev)
mip = torch.where(rough < 0.5, (rough.clamp(0.04, 0.5) - 0.04) / 0.46 * 4, (rough.clamp(0.5, 1.0) - 0.5) / 0.5 + 4)


def timeit(name, fn, bwd_inputs=None):
    for fb in ((False, True) if bwd_inputs else (False,)):
        def run():
            out = fn()
            if fb:
                torch.autograd.grad(out, bwd_inputs, torch.ones_like(out))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("%-46s %-8s %.3f ms" % (name, "fwd+bwd" if fb else "fwd", e0.elapsed_time(e1) / 20))


timeit("diffuse: cube 16^2 linear", lambda: dr.texture(diffuse, n, filter_mode="linear", boundary_mode="cube"), [diffuse])
timeit("BRDF LUT: 2-D 256^2 clamp linear", lambda: dr.texture(lut, uv, filter_mode="linear", boundary_mode="clamp"))
timeit("specular: cube 512^2 + 5 mips, mip-linear by bias", lambda: dr.texture(spec[0], n, mip=spec[1:], mip_level_bias=mip, filter_mode="linear-mipmap-linear", boundary_mode="cube"), spec)
