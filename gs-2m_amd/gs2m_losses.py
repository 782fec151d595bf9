"""The per-iteration loss terms of the reference's training loop (train.py:101-130) that are plain PyTorch there and
stay PyTorch here, restated from utils/loss_utils.py (l1_loss :27-28, plane_loss :72-79, depth_normal_loss :113-120,
_get_img_grad_weight :122-135, tv_loss :536-557).  The D-SSIM term is the `fused_ssim` package (HIP)."""
import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def plane_loss(visibility_filter, gaussians):
    """Mean of the smallest scale of every visible Gaussian (flattening prior)."""
    # masked form of `get_scaling[visibility_filter] ... .mean()`: no nonzero() / gather / index_put_ backward, and no
    # host synchronisation on the count (an empty filter gives 0 here as it does there)
    n = visibility_filter.sum()
    smallest = gaussians.get_scaling.min(dim=-1).values
    return torch.where(visibility_filter, smallest, 0.0).sum() / n.clamp(min=1)


def image_gradient_weight(img):
    """(3,H,W) -> (H,W): central-difference edge strength, min-max normalised, zero on the 1-pixel border."""
    gx = (img[:, 1:-1, 2:] - img[:, 1:-1, :-2]).abs().mean(0)
    gy = (img[:, :-2, 1:-1] - img[:, 2:, 1:-1]).abs().mean(0)
    g = torch.maximum(gx, gy)
    g = (g - g.min()) / (g.max() - g.min())
    return F.pad(g[None, None], (1, 1, 1, 1), mode="constant", value=0.0)[0, 0]


def edge_weights(gt_image):
    """(1 - normalised image gradient)^2: the per-pixel weight of depth_normal_loss.  Depends on the ground-truth image only,
    so a training loop can compute it once per view."""
    return (1.0 - image_gradient_weight(gt_image)).clamp(0, 1).detach() ** 2


def depth_normal_loss(normal_map, sobel_map, gt_image=None, weight_map=None, weights=None):
    """L1 between the rendered normals and the normals of the rendered depth, down-weighted at image edges
    (`weights`: a precomputed `edge_weights(gt_image)`)."""
    if weights is None:
        weights = edge_weights(gt_image)
    if weight_map is not None:
        weights = weights * weight_map.squeeze()
    return (weights * (sobel_map - normal_map).abs().sum(dim=0)).mean()


def tv_loss(gt_image, pred, norm1=True, weight_map=None):
    """Edge-aware total variation of `pred` (C,H,W): neighbour differences (L1, or squared with norm1=False) damped where the
    reference image (3,H,W) has an edge, optionally weighted per pixel (1,H,W)."""
    edge_h = torch.exp(-(gt_image[:, 1:, :] - gt_image[:, :-1, :]).abs().mean(dim=0, keepdim=True))
    edge_w = torch.exp(-(gt_image[:, :, 1:] - gt_image[:, :, :-1]).abs().mean(dim=0, keepdim=True))
    dh, dw = pred[:, 1:, :] - pred[:, :-1, :], pred[:, :, 1:] - pred[:, :, :-1]
    loss_h = (dh.abs() if norm1 else dh * dh) * edge_h
    loss_w = (dw.abs() if norm1 else dw * dw) * edge_w
    if weight_map is not None:
        loss_h = loss_h * ((weight_map[:, 1:, :] + weight_map[:, :-1, :]) / 2.0)
        loss_w = loss_w * ((weight_map[:, :, 1:] + weight_map[:, :, :-1]) / 2.0)
    return loss_h.mean() + loss_w.mean()
