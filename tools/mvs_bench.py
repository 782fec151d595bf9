"""Photometric multi-view term at the reference's sample count (102,400 pixels, 7x7 patches, 1920x1080 grey images):
the op-by-op PyTorch formulation (utils/loss_utils.py:303-349 as restated in gs2m_mvs.patch_ncc_torch) against the fused
kernel, forward and forward + backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import gs2m_synth as S
import gs2m_mvs as MV
from gs2m_scene import Camera

W, H, N = 1920, 1080, 102400
dev = "cuda"
ref = Camera(S.look_at_camera(W, H, (0.0, 0.0, 0.0), (0.0, 0.0, 6.0)), dev)
near = Camera(S.look_at_camera(W, H, (0.5, -0.2, 0.2), (0.0, 0.0, 6.0)), dev)
g = torch.Generator().manual_seed(0)
for c in (ref, near):
    c.gray_image = torch.nn.functional.avg_pool2d(torch.rand(1, 1, H + 8, W + 8, generator=g), 9, stride=1, padding=0)[0].to(dev).contiguous()
pixels = torch.stack([torch.rand(N, generator=g) * (W - 1), torch.rand(N, generator=g) * (H - 1)], dim=-1).to(dev)
n = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, -1.0]) + 0.1 * torch.randn(N, 3, generator=g), dim=-1).to(dev).requires_grad_(True)
d = (6.0 + 0.2 * torch.randn(N, generator=g)).to(dev).requires_grad_(True)
for name, fn in (("PyTorch op by op", MV.patch_ncc_torch), ("fused HIP kernel", MV.patch_ncc)):
    for fb in (False, True):
        def run():
            ncc, _ = fn(pixels, n, d, ref, near, 1.0, 3)
            if fb:
                torch.autograd.grad(ncc.sum(), [n, d])
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("%-20s %-8s %.3f ms" % (name, "fwd+bwd" if fb else "fwd", e0.elapsed_time(e1) / 10))
