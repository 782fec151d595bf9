"""fused_ssim (HIP, include/gs2m_ssim.h) against the reference's PyTorch formulation of the same operator
(utils/loss_utils.py:30-70: five depthwise conv2d with the 11x11 Gaussian window) -- the comparison the reference's
own submodules/fused-ssim/tests/test.py makes (`torch.isclose` on the value and on the gradient)."""
from math import exp

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _window(channel, dev):
    g = torch.Tensor([exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, 11, 11).contiguous().to(dev)


def torch_ssim_map(img1, img2, padding="same"):
    ch = img1.size(-3)
    w = _window(ch, img1.device).double()
    img1, img2 = img1.double(), img2.double()  # fp64: the arbiter both fp32 implementations are compared with
    pad = 5 if padding == "same" else 0
    mu1 = F.conv2d(img1, w, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, w, padding=pad, groups=ch)
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=ch) - mu1 * mu1
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=ch) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=ch) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


@pytest.mark.parametrize("shape", [(1, 3, 64, 64), (2, 3, 45, 64), (1, 1, 7, 9), (1, 3, 46, 129), (3, 2, 100, 63), (1, 3, 11, 200), (1, 3, 136, 75)])
@pytest.mark.parametrize("padding", ["same", "valid"])
def test_fused_ssim_matches_torch(shape, padding):
    assert torch.cuda.is_available()
    from fused_ssim import fused_ssim, FusedSSIMMap
    if padding == "valid" and min(shape[2:]) <= 10:
        pytest.skip("valid padding needs more than 10 pixels")
    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(shape, generator=gen).cuda().requires_grad_(True)
    b = torch.rand(shape, generator=gen).cuda()
    b[..., : shape[3] // 2] = (a.detach()[..., : shape[3] // 2] + 0.05 * b[..., : shape[3] // 2]).clamp(0, 1)  # a well-matched half
    ref_map = torch_ssim_map(a, b, padding)
    ref = ref_map.mean()
    ref.backward()
    g_ref = a.grad.clone()
    a.grad = None
    got_map = FusedSSIMMap.apply(0.01 ** 2, 0.03 ** 2, a, b, padding, True)
    assert got_map.shape == ref_map.shape
    assert (got_map.double() - ref_map).abs().max().item() < 2e-5   # fp32 cancellation in sigma = E[x^2] - mu^2
    got = fused_ssim(a, b, padding)
    assert torch.isclose(got.double(), ref, rtol=1e-5, atol=1e-6)
    got.backward()
    err = (a.grad.double() - g_ref).abs().max().item()
    assert err < 1e-4 * g_ref.abs().max().item() + 1e-9, err
    # a non-uniform upstream gradient through the map (FusedSSIMMap is a general autograd node)
    a.grad = None
    G = torch.rand(ref_map.shape, generator=gen).cuda()
    (torch_ssim_map(a, b, padding) * G.double()).sum().backward()
    g_ref = a.grad.clone()
    a.grad = None
    (FusedSSIMMap.apply(0.01 ** 2, 0.03 ** 2, a, b, padding, True) * G).sum().backward()
    err = (a.grad.double() - g_ref).abs().max().item()
    assert err < 1e-4 * g_ref.abs().max().item() + 1e-9, err


def test_fused_ssim_identical_images_and_inference_mode():
    assert torch.cuda.is_available()
    from fused_ssim import fused_ssim
    a = torch.rand(1, 3, 90, 130, generator=torch.Generator().manual_seed(0)).cuda()
    assert abs(fused_ssim(a, a.clone(), train=False).item() - 1.0) < 1e-5
    v = fused_ssim(a, torch.flip(a, dims=(3,)), train=False)
    assert torch.isclose(v.double(), torch_ssim_map(a, torch.flip(a, dims=(3,))).mean(), rtol=1e-5, atol=1e-6)
    x = a.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="train=True"):
        fused_ssim(x, a, train=False).backward()


def test_fused_ssim_fullsize_properties():
    """1080p: symmetric in its arguments (value), 1 on identical images, gradient of mean(ssim(a, a)) is ~0."""
    assert torch.cuda.is_available()
    from fused_ssim import fused_ssim
    gen = torch.Generator().manual_seed(1)
    a = torch.rand(1, 3, 1080, 1920, generator=gen).cuda().requires_grad_(True)
    b = torch.rand(1, 3, 1080, 1920, generator=gen).cuda()
    v1, v2 = fused_ssim(a, b), fused_ssim(b, a.detach(), train=False)
    assert torch.isclose(v1, v2, rtol=1e-6, atol=1e-7)
    ref = torch_ssim_map(a.detach()[:, :, :200, :300], b[:, :, :200, :300])[:, :, :190, :290]   # interior of a crop
    from fused_ssim import fusedssim
    got = fusedssim(0.01 ** 2, 0.03 ** 2, a.detach(), b, False)[0][:, :, :190, :290]
    assert (got.double() - ref).abs().max().item() < 2e-5
    s = fused_ssim(a, a.detach().clone())
    assert abs(s.item() - 1.0) < 1e-5
    s.backward()
    assert a.grad.abs().max().item() < 1e-6


@pytest.mark.parametrize("shape", [(1, 3, 1080, 1920), (2, 3, 45, 67), (1, 1, 7, 9)])
def test_dssim_loss_is_weight_times_one_minus_mean_ssim(shape):
    """train.py:103's term as one node: the value, and the gradient against the map formulation with the same kernels."""
    from fused_ssim import dssim_loss, FusedSSIMMap
    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(shape, generator=gen).cuda().requires_grad_(True)
    b = torch.rand(shape, generator=gen).cuda()
    ref = 0.2 * (1.0 - FusedSSIMMap.apply(0.01 ** 2, 0.03 ** 2, a, b, "same", True).mean())
    (ref * 1.7).backward()
    g_ref = a.grad.clone()
    a.grad = None
    got = dssim_loss(a, b, 0.2)
    (got * 1.7).backward()
    assert torch.isclose(got, ref.detach(), rtol=2e-6, atol=1e-7)
    assert torch.allclose(a.grad, g_ref, rtol=1e-5, atol=1e-12), (a.grad - g_ref).abs().max()
    assert torch.equal(dssim_loss(a.detach().requires_grad_(True), b, 0.2), got), "fixed-order reduction: bitwise reproducible"
