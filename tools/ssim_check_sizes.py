import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from fused_ssim import fused_ssim
from test_ssim_gpu import torch_ssim_map
g = torch.Generator().manual_seed(0)
for (B, C, H, W) in ((1, 3, 720, 1280), (1, 3, 581, 777), (2, 3, 1080, 1920), (1, 1, 2160, 3840), (1, 3, 333, 555)):
    a = torch.rand(B, C, H, W, generator=g).cuda().requires_grad_(True); b = torch.rand(B, C, H, W, generator=g).cuda()
    v = fused_ssim(a, b); (ga,) = torch.autograd.grad(v, a)
    a2 = a.detach().clone().requires_grad_(True)
    v2 = torch_ssim_map(a2, b).mean(); (gb,) = torch.autograd.grad(v2, a2)
    print((B, C, H, W), "value diff", abs(float(v) - float(v2)), "grad max rel", float((ga - gb).abs().max() / gb.abs().max()))
