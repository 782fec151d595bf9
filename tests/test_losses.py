"""gs2m_losses' PyTorch expressions against OUTPUTS OF THE REFERENCE's own functions (tests/golden/ref_losses.npz, written by
tests/golden/make_golden.py from /root/reference/utils/loss_utils.py): l1_loss, ssim, _get_img_grad_weight,
depth_normal_loss, tv_loss, plane_loss.  The fused HIP forms are checked against the same vectors in test_losses_gpu.py /
test_ssim_gpu.py."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_losses.npz")


def load():
    z = np.load(GOLD)
    return {k: torch.tensor(z[k]) if z[k].dtype != np.bool_ else torch.tensor(z[k]) for k in z.files}


def test_pytorch_expressions_reproduce_the_reference_functions():
    import gs2m_losses as L
    z = load()
    rgb = z["img"].clamp(0, 1)
    close = lambda a, b, tol=1e-6: abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))
    assert close(L.l1_loss(rgb, z["gt"]), z["l1"])
    assert torch.allclose(L.image_gradient_weight(z["gt"]), z["img_grad_weight"], rtol=1e-6, atol=1e-7)
    assert close(L.depth_normal_loss(z["normal"], z["sobel"], z["gt"]), z["depth_normal"])
    assert close(L.depth_normal_loss(z["normal"], z["sobel"], z["gt"], weight_map=z["wm"]), z["depth_normal_wm"])
    assert close(L.depth_normal_loss(z["normal"], z["sobel"], weights=L.edge_weights(z["gt"])), z["depth_normal"])
    assert close(L.tv_loss(z["gt"], z["pred1"], norm1=False), z["tv_l2_c1"])
    assert close(L.tv_loss(z["gt"], z["pred3"]), z["tv_l1_c3"])
    assert close(L.tv_loss(z["gt"], z["pred3"], weight_map=z["wm"]), z["tv_l1_c3_wm"])
    model = types.SimpleNamespace(get_scaling=torch.exp(z["raw_scale"]))
    assert close(L.plane_loss(z["vis"], model), z["plane"])
    assert float(L.plane_loss(torch.zeros_like(z["vis"]), model)) == float(z["plane_none_visible"]) == 0.0


def test_multi_view_helpers_reproduce_the_reference_functions():
    """gs2m_mvs' restatements of the pure helpers of the multi-view terms against OUTPUTS OF THE REFERENCE's own functions
    (tests/golden/ref_mvs.npz: _patch_offsets, _patch_warp, _loss_ncc in both modes, _patch_gradient, _sample_normal_map,
    _sample_depth_normal of utils/loss_utils.py) -- the formulation the fused patch-NCC / geometry kernels are tested against."""
    import gs2m_mvs as M
    z = np.load(os.path.join(os.path.dirname(GOLD), "ref_mvs.npz"))
    t = lambda k: torch.tensor(z[k])
    for h in (1, 3):
        assert torch.equal(M._patch_offsets(h, "cpu"), t(f"offsets_h{h}"))
    B = t("warp_H").shape[0]
    assert torch.allclose(M._patch_warp(t("warp_H").reshape(B, 9), t("warp_uv")), t("warp_grid"), rtol=1e-6, atol=1e-6)
    ncc, mask = M._loss_ncc(t("ncc_ref"), t("ncc_nea"))
    assert torch.allclose(ncc, t("ncc"), rtol=1e-5, atol=1e-6) and torch.equal(mask, t("ncc_mask"))
    _, smask = M._loss_ncc(t("ncc_ref"), t("ncc_nea"), std_mask=True)
    assert torch.equal(smask, t("ncc_std_mask")) and bool(smask[4].all()) and not bool(smask[5:].any())
    assert torch.allclose(M._patch_gradient(t("ncc_ref"), 7), t("patch_gradient"), rtol=1e-6, atol=1e-6)
    assert torch.allclose(M._sample_normal_map(t("sn_pixels"), t("sn_normal_map"), fused=False), t("sn_out"), rtol=1e-6, atol=1e-6)
    fx, fy, cx, cy, W, H = (float(v) for v in z["sdn_cam"])
    cam = types.SimpleNamespace(Fx=fx, Fy=fy, Cx=cx, Cy=cy, image_width=int(W), image_height=int(H))
    zz, n, valid = M._sample_depth_normal(t("sdn_pts"), cam, {"depth_map": t("sdn_depth_map"), "normal_map": t("sn_normal_map")}, fused=False)
    assert torch.equal(valid, t("sdn_valid"))
    assert torch.allclose(zz, t("sdn_z"), rtol=1e-6, atol=1e-6) and torch.allclose(n, t("sdn_n"), rtol=1e-5, atol=1e-6)


def test_random_subset_is_an_exact_uniform_draw():
    """gs2m_mvs.random_subset: exactly min(k, count) indices of set mask elements, no duplicates, every set element equally likely"""
    import gs2m_mvs
    torch.manual_seed(0)
    mask = torch.rand(20000) < 0.6
    n = int(mask.sum())
    hits = torch.zeros(20000)
    for _ in range(200):
        idx = gs2m_mvs.random_subset(mask, 3000)
        assert idx.numel() == 3000 and len(torch.unique(idx)) == 3000 and bool(mask[idx].all())
        hits[idx] += 1
    freq = hits[mask] / 200.0
    assert abs(float(freq.mean()) - 3000.0 / n) < 1e-6 and float(freq.std()) < 1.5 * (3000.0 / n * (1 - 3000.0 / n) / 200.0) ** 0.5
    few = torch.zeros(5000, dtype=torch.bool); few[::50] = True
    idx = gs2m_mvs.random_subset(few, 3000)
    assert idx.numel() == 100 and bool(few[idx].all())
    assert gs2m_mvs.random_subset(torch.zeros(100, dtype=torch.bool), 10).numel() == 0
    near = torch.ones(3100, dtype=torch.bool)  # count within 4 sigma of k: everything is kept by the thinning
    assert gs2m_mvs.random_subset(near, 3000).numel() == 3000


def test_take_distinct_matches_indexing():
    import gs2m_mvs
    torch.manual_seed(1)
    for shape in ((5000,), (5000, 3)):
        x = torch.randn(*shape, requires_grad=True)
        idx = torch.randperm(5000)[:1200]
        w = torch.randn(1200, *shape[1:])
        (ga,) = torch.autograd.grad((x[idx] * w).sum(), x)
        (gb,) = torch.autograd.grad((gs2m_mvs.take_distinct(x, idx) * w).sum(), x)
        assert torch.equal(x[idx], gs2m_mvs.take_distinct(x, idx)) and torch.equal(ga, gb)
