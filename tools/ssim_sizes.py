import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
from fused_ssim import fused_ssim
for (H, W) in ((581, 777), (1080, 1920), (360, 640)):
    a = torch.rand(1, 3, H, W, device="cuda", requires_grad=True); b = torch.rand(1, 3, H, W, device="cuda")
    for _ in range(10):
        a.grad = None; fused_ssim(a, b).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        a.grad = None; fused_ssim(a, b).backward()
    torch.cuda.synchronize(); print(f"{W}x{H}: fused_ssim forward + backward {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
