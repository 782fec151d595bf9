"""The fused loss tail (csrc/loss_ops.hip, include/gs2m_loss.h) against the PyTorch expressions of the
reference it replaces (gs2m_losses: l1_loss, depth_normal_loss + edge_weights, plane_loss; GaussianModel's statistics),
values and gradients, on the device."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
pytestmark = pytest.mark.gpu


def _frames(H, W, seed, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    image = (torch.randn(3, H, W, generator=g) * 0.45 + 0.5)          # a good share outside [0, 1]: the clamp's mask matters
    gt = torch.rand(3, H, W, generator=g)
    normal = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    sobel = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    R = torch.randn(3, H, W, generator=g)                             # a second consumer of rgb (stands in for D-SSIM)
    wm = torch.rand(1, H, W, generator=g)
    return [t.to(dev) for t in (image, gt, normal, sobel, R, wm)]


@pytest.mark.parametrize("H,W", [(1080, 1920), (77, 333), (5, 4)])
@pytest.mark.parametrize("mode", ["l1+dn", "l1+dn+weight_map", "l1", "dn unweighted"])
def test_geometry_image_loss_equals_the_pytorch_expressions(H, W, mode):
    import gs2m_losses as L
    image, gt, normal, sobel, R, wm = _frames(H, W, seed=H + W)
    w_l1, w_dn = 0.8, 0.015
    # --- PyTorch formulation (train.py:101-120)
    i0, n0, s0 = (t.clone().requires_grad_(True) for t in (image, normal, sobel))
    rgb0 = i0.clamp(0, 1)
    ref = w_l1 * L.l1_loss(rgb0, gt) + (rgb0 * R).sum() * 1e-3
    if mode != "l1":
        if mode == "dn unweighted":
            ref = ref + w_dn * (s0 - n0).abs().sum(dim=0).mean()
        else:
            ref = ref + w_dn * L.depth_normal_loss(n0, s0, gt, weight_map=wm if "weight_map" in mode else None)
    ref.backward()
    # --- fused
    i1, n1, s1 = (t.clone().requires_grad_(True) for t in (image, normal, sobel))
    dn = mode != "l1"
    rgb1, loss, terms = L.geometry_image_loss(i1, gt, n1 if dn else None, s1 if dn else None,
                                              edge=L.edge_gradient(gt) if dn and mode != "dn unweighted" else None,
                                              weight_map=wm if "weight_map" in mode else None, w_l1=w_l1, w_dn=w_dn if dn else 0.0)
    got = loss + (rgb1 * R).sum() * 1e-3
    got.backward()
    assert torch.equal(rgb1, rgb0.detach())
    assert abs(float(got.detach()) - float(ref.detach())) <= 2e-5 * abs(float(ref.detach())) + 1e-6
    assert abs(float(terms[0]) - float(L.l1_loss(rgb0, gt).detach())) <= 1e-5
    # gradients: identical expressions element by element, up to the order of two roundings
    assert torch.allclose(i1.grad, i0.grad, rtol=1e-5, atol=1e-10)
    if dn:
        assert torch.allclose(n1.grad, n0.grad, rtol=2e-5, atol=1e-12) and torch.allclose(s1.grad, s0.grad, rtol=2e-5, atol=1e-12)
        assert float(n1.grad.abs().sum()) > 0
    else:
        assert n1.grad is None and s1.grad is None


def test_edge_gradient_is_the_reference_image_gradient_weight():
    import gs2m_losses as L
    gt = _frames(270, 480, seed=5)[1]
    edge, mm = L.edge_gradient(gt)
    gx = (gt[:, 1:-1, 2:] - gt[:, 1:-1, :-2]).abs().mean(0)
    gy = (gt[:, :-2, 1:-1] - gt[:, 2:, 1:-1]).abs().mean(0)
    g = torch.maximum(gx, gy)
    assert torch.allclose(edge[1:-1, 1:-1], g, rtol=1e-6, atol=1e-8)
    assert float(edge[0].abs().sum() + edge[-1].abs().sum() + edge[:, 0].abs().sum() + edge[:, -1].abs().sum()) == 0.0
    assert abs(float(mm[0]) - float(g.min())) <= 1e-7 and abs(float(mm[1]) - float(g.max())) <= 1e-7
    w = (1.0 - (edge - mm[0]) / (mm[1] - mm[0])).clamp(0, 1) ** 2
    assert torch.allclose(w[1:-1, 1:-1], L.edge_weights(gt)[1:-1, 1:-1], rtol=1e-5, atol=1e-6)


def test_reductions_are_bitwise_reproducible_and_leave_the_workspace_clean():
    import gs2m_losses as L
    image, gt, normal, sobel, _, _ = _frames(1080, 1920, seed=1)
    edge = L.edge_gradient(gt)
    a = [L.geometry_image_loss(image, gt, normal, sobel, edge=edge, w_l1=0.8, w_dn=0.015)[1].clone() for _ in range(3)]
    assert torch.equal(a[0], a[1]) and torch.equal(a[1], a[2])
    e2 = L.edge_gradient(gt)
    assert torch.equal(edge[1], e2[1]) and torch.equal(edge[0], e2[0])
    torch.cuda.synchronize()
    assert all(float(ws[-16:].abs().sum()) == 0.0 for ws in L._workspaces.values()), "the ticket word (behind the partial sums) is back at 0"


class _Scales:
    def __init__(self, s):
        self._s = s

    @property
    def get_scaling(self):
        return torch.exp(self._s)


class _RawScales(_Scales):  # the reference model's attribute name: the fused form reads the log-scales directly
    @property
    def _scaling(self):
        return self._s


@pytest.mark.parametrize("model", [_Scales, _RawScales])
@pytest.mark.parametrize("P,frac", [(1_000_000, 0.85), (1000, 0.5), (257, 0.0), (0, 0.0)])
def test_fused_plane_loss_equals_plane_loss(P, frac, model):
    import gs2m_losses as L
    g = torch.Generator().manual_seed(P + 1)
    raw = (torch.randn(P, 3, generator=g) - 4.0).cuda()
    vis = (torch.rand(P, generator=g) < frac).cuda()
    r0, r1 = raw.clone().requires_grad_(True), raw.clone().requires_grad_(True)
    lam = 0.01 if P == 1000 else 1.0
    a = lam * L.plane_loss(vis, _Scales(r0))
    b = L.fused_plane_loss(vis, model(r1), weight=lam)
    assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a)) + 1e-12
    (a * 3.0).backward()
    (b * 3.0).backward()
    assert torch.allclose(r1.grad, r0.grad, rtol=1e-5, atol=1e-14)
    if frac == 0.0 and P:
        assert float(b) == 0.0 and float(r1.grad.abs().sum()) == 0.0


def test_densification_stats_equal_the_model_expressions():
    import gs2m_losses as L
    P = 300_001
    g = torch.Generator().manual_seed(3)
    vg = torch.randn(P, 4, generator=g).cuda()
    vis = (torch.rand(P, generator=g) < 0.8).cuda()
    observe = torch.randint(0, 3, (P,), generator=g, dtype=torch.int32).cuda()
    radii = torch.randint(0, 200, (P,), generator=g, dtype=torch.int32).cuda()
    acc, acc_abs, den = (torch.rand(P, 1, generator=g).cuda() for _ in range(3))
    mr = (torch.rand(P, generator=g) * 100).cuda()
    f = vis[:, None]
    ref_acc = acc + torch.where(f, torch.norm(vg[:, :2], dim=-1, keepdim=True), 0.0)
    ref_abs = acc_abs + torch.where(f, torch.norm(vg[:, 2:], dim=-1, keepdim=True), 0.0)
    ref_den = den + f
    mask = (observe > 0) & vis
    ref_mr = torch.where(mask, torch.max(mr, radii), mr)
    L.densification_stats(vg, vis, acc, acc_abs, den, observe, radii, mr)
    assert torch.equal(den, ref_den) and torch.equal(mr, ref_mr)
    assert torch.allclose(acc, ref_acc, rtol=3e-7, atol=0) and torch.allclose(acc_abs, ref_abs, rtol=3e-7, atol=0)
    # without the radius update
    acc2, abs2, den2 = ref_acc.clone(), ref_abs.clone(), ref_den.clone()
    L.densification_stats(vg, vis, acc2, abs2, den2)
    assert torch.equal(den2, ref_den + f)


def test_the_fused_forms_refuse_cpu_tensors():
    import gs2m_losses as L
    x = torch.rand(3, 8, 8)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        L.geometry_image_loss(x, x)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        L.edge_gradient(x)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        L.fused_plane_loss(torch.ones(4, dtype=torch.bool), _Scales(torch.zeros(4, 3)))


@pytest.mark.parametrize("H,W", [(1080, 1920), (7, 9), (2, 2)])
@pytest.mark.parametrize("C,norm1,weighted", [(1, False, False), (3, True, False), (3, True, True), (2, False, True)])
def test_fused_tv_loss_equals_tv_loss(H, W, C, norm1, weighted):
    """The smoothness terms of the material stage (train.py:160-175; utils/loss_utils.py:536-557)."""
    import gs2m_losses as L
    g = torch.Generator().manual_seed(H * 7 + W + C)
    gt = torch.rand(3, H, W, generator=g).cuda()
    pred = torch.rand(C, H, W, generator=g).cuda()
    wm = torch.rand(1, H, W, generator=g).cuda() if weighted else None
    p0, p1 = pred.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    lam = 0.37 if weighted else 1.0   # the term's multiplier folded into the node
    a = lam * L.tv_loss(gt, p0, norm1=norm1, weight_map=wm)
    b = L.fused_tv_loss(gt, p1, norm1=norm1, weight_map=wm, weight=lam)
    assert abs(float(a.detach()) - float(b.detach())) <= 2e-5 * abs(float(a.detach())) + 1e-9
    (a * 0.7).backward()
    (b * 0.7).backward()
    assert torch.allclose(p1.grad, p0.grad, rtol=2e-5, atol=1e-12), (p1.grad - p0.grad).abs().max()
    assert torch.equal(L.fused_tv_loss(gt, pred, norm1=norm1, weight_map=wm, weight=lam), b.detach()), "bitwise reproducible"


@pytest.mark.parametrize("H,W", [(540, 960), (33, 17)])
def test_geometry_image_loss_on_the_shaded_image_of_the_material_stage(H, W):
    """train.py:141-146: where(normal_mask, clamp(render_rgb^T, 0, 1), bg) -> L1 (+ the depth-normal term), with the shading's
    (H,W,3) output handed over as it is."""
    import gs2m_losses as L
    image, gt, normal, sobel, R, _ = _frames(H, W, seed=H - W)
    hwc = image.permute(1, 2, 0).contiguous()
    mask = (torch.rand(1, H, W, generator=torch.Generator().manual_seed(3)) < 0.7).cuda()
    bg = torch.tensor([0.2, 1.0, 0.0], device="cuda")
    x0, n0, s0 = hwc.clone().requires_grad_(True), normal.clone().requires_grad_(True), sobel.clone().requires_grad_(True)
    pbr0 = torch.where(mask, x0.permute(2, 0, 1).clamp(0, 1), bg[:, None, None])
    ref = 0.8 * L.l1_loss(pbr0, gt) + 0.015 * L.depth_normal_loss(n0, s0, gt) + (pbr0 * R).sum() * 1e-3
    ref.backward()
    x1, n1, s1 = hwc.clone().requires_grad_(True), normal.clone().requires_grad_(True), sobel.clone().requires_grad_(True)
    pbr1, loss, _ = L.geometry_image_loss(x1, gt, n1, s1, edge=L.edge_gradient(gt), w_l1=0.8, w_dn=0.015, mask=mask, background=bg)
    (loss + (pbr1 * R).sum() * 1e-3).backward()
    assert torch.equal(pbr1, pbr0.detach())
    assert abs(float(loss.detach()) + float(((pbr1 * R).sum() * 1e-3).detach()) - float(ref.detach())) <= 2e-5 * abs(float(ref.detach()))
    assert x1.grad.shape == (H, W, 3) and torch.allclose(x1.grad, x0.grad, rtol=1e-5, atol=1e-10)
    assert torch.allclose(n1.grad, n0.grad, rtol=2e-5, atol=1e-12) and torch.allclose(s1.grad, s0.grad, rtol=2e-5, atol=1e-12)
    assert float(x1.grad[~mask[0]].abs().sum()) == 0.0


def test_fused_forms_reproduce_the_reference_functions():
    """The fused kernels against OUTPUTS OF THE REFERENCE's own loss functions (tests/golden/ref_losses.npz, generated from
    /root/reference/utils/loss_utils.py by tests/golden/make_golden.py): the pinned vectors of this row."""
    import types
    import gs2m_losses as L
    from fused_ssim import dssim_loss, fused_ssim
    z = {k: torch.tensor(v).cuda() for k, v in np.load(os.path.join(ROOT, "tests", "golden", "ref_losses.npz")).items()}
    val = lambda t: float(t.detach()) if torch.is_tensor(t) else float(t)
    close = lambda a, b, tol=2e-5: abs(val(a) - val(b)) <= tol * max(1.0, abs(val(b)))
    edge = L.edge_gradient(z["gt"])
    rgb, loss, terms = L.geometry_image_loss(z["img"], z["gt"], z["normal"], z["sobel"], edge=edge, w_l1=1.0, w_dn=1.0)
    assert close(terms[0], z["l1"]) and close(terms[1], z["depth_normal"]) and close(loss, z["l1"] + z["depth_normal"])
    assert torch.equal(rgb, z["img"].clamp(0, 1))
    _, _, t2 = L.geometry_image_loss(z["img"], z["gt"], z["normal"], z["sobel"], edge=edge, weight_map=z["wm"], w_l1=0.0, w_dn=1.0)
    assert close(t2[1], z["depth_normal_wm"])
    w = (1.0 - (edge[0] - edge[1][0]) / (edge[1][1] - edge[1][0])).clamp(0, 1)
    assert torch.allclose((1.0 - w)[1:-1, 1:-1], z["img_grad_weight"][1:-1, 1:-1], rtol=1e-5, atol=1e-6)
    assert close(L.fused_tv_loss(z["gt"], z["pred1"], norm1=False), z["tv_l2_c1"])
    assert close(L.fused_tv_loss(z["gt"], z["pred3"]), z["tv_l1_c3"])
    assert close(L.fused_tv_loss(z["gt"], z["pred3"], weight_map=z["wm"]), z["tv_l1_c3_wm"])
    raw = z["raw_scale"]
    assert close(L.fused_plane_loss(z["vis"], types.SimpleNamespace(_scaling=raw, get_scaling=torch.exp(raw))), z["plane"])
    assert float(L.fused_plane_loss(torch.zeros_like(z["vis"]), types.SimpleNamespace(get_scaling=torch.exp(raw)))) == 0.0
    assert close(fused_ssim(rgb.unsqueeze(0), z["gt"].unsqueeze(0), train=False), z["ssim"], 1e-4)
    assert close(dssim_loss(rgb.unsqueeze(0).requires_grad_(True), z["gt"].unsqueeze(0), 0.2), 0.2 * (1.0 - float(z["ssim"])), 1e-4)
