"""The per-iteration loss terms of the reference's training loop (train.py:101-130) that are plain PyTorch there and
stay PyTorch here, restated from utils/loss_utils.py (l1_loss :27-28, plane_loss :72-79, depth_normal_loss :113-120,
_get_img_grad_weight :122-135).  The D-SSIM term is the `fused_ssim` package (HIP)."""
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


def depth_normal_loss(normal_map, sobel_map, gt_image, weight_map=None):
    """L1 between the rendered normals and the normals of the rendered depth, down-weighted at image edges."""
    weights = (1.0 - image_gradient_weight(gt_image)).clamp(0, 1).detach() ** 2
    if weight_map is not None:
        weights = weights * weight_map.squeeze()
    return (weights * (sobel_map - normal_map).abs().sum(dim=0)).mean()
