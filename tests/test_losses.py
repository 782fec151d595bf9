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
