"""The per-iteration loss terms of the reference's training loop (train.py:101-130), restated from utils/loss_utils.py
(l1_loss :27-28, plane_loss :72-79, depth_normal_loss :113-120, _get_img_grad_weight :122-135, tv_loss :536-557), twice:
  * as the plain PyTorch expressions the reference runs (`l1_loss`, `plane_loss`, `depth_normal_loss`, ...): any device,
    the formulation the fused kernels are tested against;
  * fused for MI355X (`geometry_image_loss`, `fused_plane_loss`, `edge_gradient`, `densification_stats`): hand-written HIP
    behind the C ABI of include/gs2m_loss.h (csrc/loss_ops.hip), CUDA tensors only -- ~90 small framework kernels per
    iteration become 6 launches.  No CPU path: a CPU tensor raises.
The D-SSIM term is the `fused_ssim` package (HIP)."""
import ctypes as C

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


# ---------------------------------------------------------------------------------------------------------------------
# fused forms (csrc/loss_ops.hip)
_workspaces = {}


def _native():
    import gs2m_native
    return gs2m_native


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device):
    """raw hipStream_t (int) of torch's current stream on `device`"""
    return _native().stream_ptr(device)


def _workspace(device):
    """The reducing kernels' scratch (partial sums + ticket): zero when created, left zeroed by every call; one per
    (device, stream) because concurrent launches on two streams must not share it."""
    st = _stream(device)
    key = (device.index, st)
    ws = _workspaces.get(key)
    if ws is None:
        # zero-filled on the current stream: the one that will use it
        ws = torch.zeros(_native().lib().gs2m_loss_workspace_bytes() // 4, dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def _cuda_f32(t, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"gs2m_losses: {name} must be a CUDA tensor (the fused losses are HIP kernels; use the PyTorch forms on the CPU)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"gs2m_losses: {name} must be float32")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"gs2m_losses: {name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


def edge_gradient(gt_image):
    """(3,H,W) ground truth -> (edge (H,W), minmax (2,)): the un-normalised image-gradient strength of
    `image_gradient_weight` and its interior min / max -- what `geometry_image_loss` turns into the per-pixel weight of
    the depth-normal term.  Depends on the ground truth only: a training loop computes it once per view."""
    gt = _cuda_f32(gt_image, "gt_image")
    _, H, W = gt.shape
    edge = torch.empty(H, W, dtype=torch.float32, device=gt.device)
    minmax = torch.empty(2, dtype=torch.float32, device=gt.device)
    with _native().device_guard(gt.device):
        _native().check(_native().lib().gs2m_edge_gradient(W, H, _ptr(gt), _ptr(edge), _ptr(minmax), _ptr(_workspace(gt.device)),
                                                           C.c_void_p(_stream(gt.device))), "gs2m_edge_gradient")
    return edge, minmax


class _GeometryImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, normal_map, sobel_map, edge, minmax, weight_map, w_l1, w_dn, mask, background):
        image = _cuda_f32(image, "image")
        gt = _cuda_f32(gt, "gt")
        _, H, W = gt.shape
        if tuple(image.shape) == (3, H, W):
            hwc = 0
        elif tuple(image.shape) == (H, W, 3):
            hwc = 1
        else:
            raise RuntimeError(f"gs2m_losses: image must be (3, {H}, {W}) or ({H}, {W}, 3), got {tuple(image.shape)}")
        if mask is not None:
            if mask.dtype != torch.bool or mask.numel() != H * W or mask.device != image.device:
                raise RuntimeError("gs2m_losses: mask must be a bool tensor of H x W elements on the image's device")
            mask = mask.contiguous()
            background = _cuda_f32(background, "background", (3,))
        dn = normal_map is not None
        if dn:
            normal_map, sobel_map = _cuda_f32(normal_map, "normal_map", (3, H, W)), _cuda_f32(sobel_map, "sobel_map", (3, H, W))
        if edge is not None:
            edge, minmax = _cuda_f32(edge, "edge", (H, W)), _cuda_f32(minmax, "minmax", (2,))
        if weight_map is not None:
            weight_map = _cuda_f32(weight_map.reshape(H, W), "weight_map")
        rgb = torch.empty_like(gt)
        out = torch.empty(3, dtype=torch.float32, device=image.device)
        with _native().device_guard(image.device):
            _native().check(_native().lib().gs2m_image_loss_forward(
                W, H, _ptr(image), hwc, _ptr(mask), _ptr(background if mask is not None else None), _ptr(gt), _ptr(normal_map), _ptr(sobel_map), _ptr(edge), _ptr(minmax), _ptr(weight_map),
                float(w_l1), float(w_dn), _ptr(rgb), _ptr(out), _ptr(_workspace(image.device)),
                C.c_void_p(_stream(image.device))), "gs2m_image_loss_forward")
        ctx.save_for_backward(image, gt, normal_map, sobel_map, edge, minmax, weight_map, mask)
        ctx.w = (float(w_l1), float(w_dn))
        ctx.hwc = hwc
        terms = out[1:]
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)  # an unused rgb / loss arrives as None (NULL for the kernel), not as a zero frame
        return rgb, out[0], terms

    @staticmethod
    def backward(ctx, g_rgb, g_loss, _g_terms):
        image, gt, normal_map, sobel_map, edge, minmax, weight_map, mask = ctx.saved_tensors
        _, H, W = gt.shape
        if g_rgb is not None:
            g_rgb = g_rgb.contiguous()
        if g_loss is not None:
            g_loss = g_loss.contiguous()
        d_image = torch.empty_like(image)
        d_normal = torch.empty_like(normal_map) if normal_map is not None else None
        d_sobel = torch.empty_like(sobel_map) if sobel_map is not None else None
        with _native().device_guard(image.device):
            _native().check(_native().lib().gs2m_image_loss_backward(
                W, H, _ptr(image), ctx.hwc, _ptr(mask), _ptr(gt), _ptr(normal_map), _ptr(sobel_map), _ptr(edge), _ptr(minmax), _ptr(weight_map),
                ctx.w[0], ctx.w[1], _ptr(g_loss), _ptr(g_rgb), _ptr(d_image), _ptr(d_normal), _ptr(d_sobel),
                C.c_void_p(_stream(image.device))), "gs2m_image_loss_backward")
        return d_image, None, d_normal, d_sobel, None, None, None, None, None, None, None


def geometry_image_loss(image, gt, normal_map=None, sobel_map=None, edge=None, weight_map=None, w_l1=1.0, w_dn=0.0, mask=None,
                        background=None):
    """train.py:101-104 and :113-120 in one pass over the frame: clamps the rendered `image` (3,H,W) to [0, 1] and returns
    `(rgb, loss, terms)` with rgb the clamped image (input of the D-SSIM term), loss = w_l1 * l1_loss(rgb, gt) +
    w_dn * depth_normal_loss(normal_map, sobel_map, ...) as a 0-dim tensor and terms = (l1, dn) detached, for logging.
    `edge` = `edge_gradient(gt)` (None: unweighted), `weight_map` (1,H,W) or (H,W) an extra per-pixel factor.  Gradients
    flow to image (through the clamp, including what arrives at rgb), normal_map and sobel_map.
    The material stage's shaded image (train.py:141-146) goes in as pbr_shading returns it: `image` (H,W,3) with
    `mask` = normal_mask and `background`: rgb = where(mask, clamp(image^T, 0, 1), background)."""
    e, mm = edge if edge is not None else (None, None)
    return _GeometryImageLoss.apply(image, gt, normal_map, sobel_map, e, mm, weight_map, w_l1, w_dn, mask, background)


class _PlaneLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, visible, raw, weight):
        scaling = _cuda_f32(scaling, "scaling")
        P = scaling.shape[0]
        if scaling.dim() != 2 or scaling.shape[1] != 3 or visible.dtype != torch.bool or visible.shape != (P,) or visible.device != scaling.device:
            raise RuntimeError("gs2m_losses: fused_plane_loss takes scaling (P,3) float32 and visibility_filter (P,) bool on one device")
        visible = visible.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=scaling.device)
        with _native().device_guard(scaling.device):
            _native().check(_native().lib().gs2m_plane_loss_forward(P, _ptr(scaling), int(raw), _ptr(visible), float(weight), _ptr(out), _ptr(_workspace(scaling.device)),
                                                                    C.c_void_p(_stream(scaling.device))), "gs2m_plane_loss_forward")
        ctx.save_for_backward(scaling, visible, out)
        ctx.raw, ctx.weight = int(raw), float(weight)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        scaling, visible, out = ctx.saved_tensors
        d = torch.empty_like(scaling)
        with _native().device_guard(scaling.device):
            _native().check(_native().lib().gs2m_plane_loss_backward(scaling.shape[0], _ptr(scaling), ctx.raw, _ptr(visible), ctx.weight, _ptr(out), _ptr(g.contiguous()),
                                                                     _ptr(d), C.c_void_p(_stream(scaling.device))), "gs2m_plane_loss_backward")
        return d, None, None, None


def fused_plane_loss(visibility_filter, gaussians, weight=1.0):
    """`weight * plane_loss(...)` as one launch each way (same arguments; `weight` = lambda_plane folded into the node).  A model that stores log-scales under the reference's name
    `_scaling` (scene/gaussian_model.py:113-114: get_scaling = exp(_scaling)) is read there directly: the exp and its
    derivative happen inside the two kernels instead of as framework ops around them."""
    raw = getattr(gaussians, "_scaling", None)
    if torch.is_tensor(raw) and getattr(gaussians, "scaling_activation", torch.exp) is torch.exp:
        return _PlaneLoss.apply(raw, visibility_filter, True, weight)
    return _PlaneLoss.apply(gaussians.get_scaling, visibility_filter, False, weight)


class _TvLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, pred, weight_map, norm1, weight):
        pred = _cuda_f32(pred, "pred")
        C, H, W = pred.shape
        gt = _cuda_f32(gt, "gt_image", (3, H, W))
        if weight_map is not None:
            weight_map = _cuda_f32(weight_map.reshape(H, W), "weight_map")
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        with _native().device_guard(pred.device):
            _native().check(_native().lib().gs2m_tv_loss_forward(W, H, C, _ptr(gt), _ptr(pred), _ptr(weight_map), int(bool(norm1)), float(weight), _ptr(out),
                                                                 _ptr(_workspace(pred.device)), C_void(_stream(pred.device))), "gs2m_tv_loss_forward")
        ctx.save_for_backward(gt, pred, weight_map)
        ctx.norm1, ctx.weight = int(bool(norm1)), float(weight)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        gt, pred, weight_map = ctx.saved_tensors
        C, H, W = pred.shape
        d = torch.empty_like(pred)
        with _native().device_guard(pred.device):
            _native().check(_native().lib().gs2m_tv_loss_backward(W, H, C, _ptr(gt), _ptr(pred), _ptr(weight_map), ctx.norm1, ctx.weight, _ptr(g.contiguous()),
                                                                  _ptr(d), C_void(_stream(pred.device))), "gs2m_tv_loss_backward")
        return None, d, None, None, None


def C_void(stream):
    return C.c_void_p(stream)


def fused_tv_loss(gt_image, pred, norm1=True, weight_map=None, weight=1.0):
    """`weight * tv_loss(...)` (same arguments; `weight` = the term's lambda folded into the node) as one launch each way; the
    gradient goes to `pred` only (the loop passes a detached weight map and the ground truth)."""
    return _TvLoss.apply(gt_image, pred, weight_map, norm1, weight)


def densification_stats(viewspace_grad, visibility_filter, grad_accum, grad_accum_abs, denom, observe=None, radii=None, max_radii=None):
    """scene/gaussian_model.py:569-573 (+ train.py:223-225 when observe / radii / max_radii are given) in place, one launch."""
    vg = _cuda_f32(viewspace_grad, "viewspace_grad")
    P = vg.shape[0]
    if tuple(vg.shape) != (P, 4):
        raise RuntimeError("gs2m_losses: viewspace_grad must be (P, 4)")
    for t, n in ((grad_accum, "grad_accum"), (grad_accum_abs, "grad_accum_abs"), (denom, "denom")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == P):
            raise RuntimeError(f"gs2m_losses: {n} must be a contiguous float32 CUDA tensor of P elements (updated in place)")
    vis = visibility_filter.contiguous()
    if vis.dtype != torch.bool or vis.numel() != P:
        raise RuntimeError("gs2m_losses: visibility_filter must be (P,) bool")
    if max_radii is not None:
        if not (max_radii.is_cuda and max_radii.dtype == torch.float32 and max_radii.is_contiguous() and max_radii.numel() == P):
            raise RuntimeError("gs2m_losses: max_radii must be a contiguous float32 CUDA tensor of P elements (updated in place)")
        observe, radii = observe.contiguous(), radii.contiguous()
        if observe.dtype != torch.int32 or radii.dtype != torch.int32 or observe.numel() != P or radii.numel() != P:
            raise RuntimeError("gs2m_losses: observe and radii must be (P,) int32")
    with _native().device_guard(vg.device):
        _native().check(_native().lib().gs2m_densification_stats(
            P, _ptr(vg), _ptr(vis), _ptr(observe if max_radii is not None else None), _ptr(radii if max_radii is not None else None),
            _ptr(grad_accum), _ptr(grad_accum_abs), _ptr(denom), _ptr(max_radii), C.c_void_p(_stream(vg.device))), "gs2m_densification_stats")
