"""Drop-in for the reference's `fused_ssim` package (submodules/fused-ssim/fused_ssim/__init__.py:1-41) on MI355X.

Same names and argument meaning: `fused_ssim(img1, img2, padding="same", train=True)` returns the mean SSIM over
the (B, CH, H, W) map (11x11 Gaussian window, sigma 1.5, zero padding; "valid" crops the 5-pixel border), with a
gradient for `img1` only; `FusedSSIMMap` is the autograd Function; `fusedssim` / `fusedssim_backward` are the two
extension entry points (`fused_ssim_cuda` in the reference).  The kernels are hand-written HIP behind the C ABI of
include/gs2m_ssim.h (csrc/ssim.hip).  There is no CPU path.
"""
import ctypes as C

import torch

import gs2m_native as _native

allowed_padding = ["same", "valid"]


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check(t, name, like=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"fused_ssim: {name} must be a CUDA tensor (the op is a HIP kernel; there is no CPU path)")
    if t.dtype != torch.float32 or t.dim() != 4:
        raise RuntimeError(f"fused_ssim: {name} must be a float32 (B, CH, H, W) tensor")
    if like is not None and (t.shape != like.shape or t.device != like.device):
        raise RuntimeError(f"fused_ssim: {name} must have the shape and device of img1")
    return t.contiguous()


def fusedssim(C1, C2, img1, img2, train):
    """-> (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12); the last three are empty tensors when train is False
    (submodules/fused-ssim/ssim.h:7-14)."""
    img1 = _check(img1, "img1")
    img2 = _check(img2, "img2", img1)
    B, CH, H, W = img1.shape
    ssim_map = torch.empty_like(img1)
    if train:
        d = torch.empty((3,) + tuple(img1.shape), dtype=torch.float32, device=img1.device)
        dm = (d[0], d[1], d[2])
    else:
        dm = tuple(torch.empty(0, dtype=torch.float32, device=img1.device) for _ in range(3))
    with _native.device_guard(img1.device):
        _native.check(_native.lib().gs2m_ssim_forward(
            B, CH, H, W, float(C1), float(C2), _ptr(img1), _ptr(img2), _ptr(ssim_map),
            *( [_ptr(t) for t in dm] if train else [None, None, None]),
            C.c_void_p(_native.stream_ptr(img1.device))), "gs2m_ssim_forward")
    return (ssim_map,) + dm


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    """-> dL/dimg1 (submodules/fused-ssim/ssim.h:16-26)."""
    img1 = _check(img1, "img1")
    img2 = _check(img2, "img2", img1)
    dL_dmap = _check(dL_dmap, "dL_dmap", img1)
    maps = [_check(t, n, img1) for t, n in ((dm_dmu1, "dm_dmu1"), (dm_dsigma1_sq, "dm_dsigma1_sq"), (dm_dsigma12, "dm_dsigma12"))]
    B, CH, H, W = img1.shape
    grad = torch.empty_like(img1)
    with _native.device_guard(img1.device):
        _native.check(_native.lib().gs2m_ssim_backward(
            B, CH, H, W, _ptr(img1), _ptr(img2), _ptr(dL_dmap), _ptr(maps[0]), _ptr(maps[1]), _ptr(maps[2]), _ptr(grad),
            C.c_void_p(_native.stream_ptr(img1.device))), "gs2m_ssim_backward")
    return grad


class FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = fusedssim(C1, C2, img1, img2, train)
        if padding == "valid":
            ssim_map = ssim_map[:, :, 5:-5, 5:-5]
        ctx.save_for_backward(img1.detach(), img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        ctx.C1, ctx.C2, ctx.padding, ctx.train = C1, C2, padding, train
        return ssim_map

    @staticmethod
    def backward(ctx, opt_grad):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        if not ctx.train:
            raise RuntimeError("fused_ssim: backward needs the derivative maps; call with train=True")
        dL_dmap = opt_grad
        if ctx.padding == "valid":
            dL_dmap = torch.zeros_like(img1)
            dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
        grad = fusedssim_backward(ctx.C1, ctx.C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        return None, None, grad, None, None, None


class _FusedSSIMAffineMean(torch.autograd.Function):
    """a + b * mean(ssim_map) as ONE autograd node ("same" padding): the mean is a fixed-order reduction kernel
    (gs2m_affine_mean) and the backward hands the scalar upstream gradient straight to the SSIM backward kernel
    (gs2m_ssim_backward_uniform) -- the map's gradient (a constant) is never materialised."""

    @staticmethod
    def forward(ctx, C1, C2, img1, img2, a, b):
        import gs2m_losses
        ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = fusedssim(C1, C2, img1, img2, True)
        out = torch.empty(1, dtype=torch.float32, device=img1.device)
        with _native.device_guard(img1.device):
            _native.check(_native.lib().gs2m_affine_mean(ssim_map.numel(), _ptr(ssim_map), float(a), float(b), _ptr(out),
                                                         _ptr(gs2m_losses._workspace(img1.device)),
                                                         C.c_void_p(_native.stream_ptr(img1.device))), "gs2m_affine_mean")
        ctx.save_for_backward(img1.detach().contiguous(), img2.contiguous(), dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        ctx.b = float(b)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        B, CH, H, W = img1.shape
        grad = torch.empty_like(img1)
        with _native.device_guard(img1.device):
            _native.check(_native.lib().gs2m_ssim_backward_uniform(
                B, CH, H, W, _ptr(img1), _ptr(img2), _ptr(g.contiguous()), ctx.b, float(img1.numel()), _ptr(dm_dmu1), _ptr(dm_dsigma1_sq),
                _ptr(dm_dsigma12), _ptr(grad), C.c_void_p(_native.stream_ptr(img1.device))), "gs2m_ssim_backward_uniform")
        return None, None, grad, None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    assert padding in allowed_padding
    if padding == "same" and train and torch.is_grad_enabled() and isinstance(img1, torch.Tensor) and img1.requires_grad:
        return _FusedSSIMAffineMean.apply(C1, C2, _check(img1, "img1"), _check(img2, "img2", img1), 0.0, 1.0)
    return FusedSSIMMap.apply(C1, C2, img1, img2, padding, train).mean()


def dssim_loss(img1, img2, weight=1.0):
    """The D-SSIM term of train.py:103 / :136, weight * (1 - fused_ssim(img1, img2)), as one autograd node (no scalar
    arithmetic kernels around the mean, no materialised map gradient)."""
    return _FusedSSIMAffineMean.apply(0.01 ** 2, 0.03 ** 2, _check(img1, "img1"), _check(img2, "img2", img1), float(weight), -float(weight))
