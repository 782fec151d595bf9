"""D-SSIM term at 1080p: the reference's PyTorch `ssim` (utils/loss_utils.py:30-70 formulation) against fused_ssim."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from fused_ssim import fused_ssim
from test_ssim_gpu import _window

a = torch.rand(1, 3, 1080, 1920, device="cuda", requires_grad=True)
b = torch.rand(1, 3, 1080, 1920, device="cuda")
w = _window(3, "cuda")


def torch_ssim(img1, img2):
    mu1, mu2 = F.conv2d(img1, w, padding=5, groups=3), F.conv2d(img2, w, padding=5, groups=3)
    s1 = F.conv2d(img1 * img1, w, padding=5, groups=3) - mu1.pow(2)
    s2 = F.conv2d(img2 * img2, w, padding=5, groups=3) - mu2.pow(2)
    s12 = F.conv2d(img1 * img2, w, padding=5, groups=3) - mu1 * mu2
    return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1.pow(2) + mu2.pow(2) + 1e-4) * (s1 + s2 + 9e-4))).mean()


for name, fn in (("torch ssim (reference formulation)", torch_ssim), ("fused_ssim (HIP)", fused_ssim)):
    for fb in (False, True):
        for _ in range(5):
            v = fn(a, b)
            if fb:
                a.grad = None
                v.backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            v = fn(a, b)
            if fb:
                a.grad = None
                v.backward()
        e1.record()
        torch.cuda.synchronize()
        print("%-36s %-10s %.3f ms" % (name, "fwd+bwd" if fb else "fwd", e0.elapsed_time(e1) / 20))
