"""`dr.texture` for the deferred PBR stage on MI355X: the argument surface of
nvdiffrast.torch.texture (submodules/nvdiffrast/nvdiffrast/torch/ops.py:427-521) restricted to the modes the reference
uses -- everything else raises NotImplementedError instead of silently doing something different:

    texture(tex (1,6,w,w,C), uv (N,h,w',3), filter_mode='linear', boundary_mode='cube')
    texture(tex (1,6,w,w,C), uv (N,h,w',3), mip=[(1,6,w/2,w/2,C), ...], mip_level_bias=(N,h,w'),
            filter_mode='linear-mipmap-linear', boundary_mode='cube')            # no uv_da: level = clamped bias
    texture(tex (1,H,W,C),   uv (N,h,w',2), filter_mode='linear', boundary_mode='clamp')

Gradients flow to `tex` and to every tensor in `mip`; `uv` and `mip_level_bias` must not require grad (the reference
detaches what feeds them, pbr/__init__.py:25-43).  HIP kernels behind include/gs2m_texture.h; no CPU path.
"""
import ctypes as C

import torch

import gs2m_native as _native


def _f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"nvdiffrast(texture): `{name}` must be a CUDA tensor (HIP kernel; there is no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"nvdiffrast(texture): `{name}` must be float32")
    return t.contiguous()


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _arr(ptrs):
    return (C.c_void_p * len(ptrs))(*ptrs)


class _CubeTexture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, bias, *levels):
        n = uv.numel() // 3
        ch = levels[0].shape[-1]
        widths = [int(l.shape[2]) for l in levels]
        out = torch.empty(uv.shape[:-1] + (ch,), dtype=torch.float32, device=uv.device)
        with torch.cuda.device(uv.device):
            _native.check(_native.lib().gs2m_texture_cube_forward(
                n, ch, len(levels), _arr([l.data_ptr() for l in levels]), (C.c_int * len(levels))(*widths), uv.data_ptr(),
                None if bias is None else bias.data_ptr(), out.data_ptr(), _stream(uv.device)), "gs2m_texture_cube_forward")
        ctx.save_for_backward(uv, bias)
        ctx.shapes, ctx.widths = [l.shape for l in levels], widths
        return out

    @staticmethod
    def backward(ctx, dy):
        uv, bias = ctx.saved_tensors
        dy = _f32(dy, "grad")
        grads = [torch.zeros(s, dtype=torch.float32, device=uv.device) for s in ctx.shapes]
        with torch.cuda.device(uv.device):
            _native.check(_native.lib().gs2m_texture_cube_backward(
                uv.numel() // 3, ctx.shapes[0][-1], len(grads), _arr([g.data_ptr() for g in grads]),
                (C.c_int * len(grads))(*ctx.widths), uv.data_ptr(), None if bias is None else bias.data_ptr(), dy.data_ptr(),
                int(uv.shape[-2]), _stream(uv.device)), "gs2m_texture_cube_backward")
        return (None, None) + tuple(grads)


class _Texture2DClamp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, tex):
        _, H, W, ch = tex.shape
        out = torch.empty(uv.shape[:-1] + (ch,), dtype=torch.float32, device=uv.device)
        with torch.cuda.device(uv.device):
            _native.check(_native.lib().gs2m_texture_2d_clamp_forward(uv.numel() // 2, ch, W, H, tex.data_ptr(), uv.data_ptr(),
                                                                      out.data_ptr(), _stream(uv.device)), "gs2m_texture_2d_clamp_forward")
        ctx.save_for_backward(uv)
        ctx.shape = tex.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        (uv,) = ctx.saved_tensors
        dy = _f32(dy, "grad")
        _, H, W, ch = ctx.shape
        g = torch.zeros(ctx.shape, dtype=torch.float32, device=uv.device)
        with torch.cuda.device(uv.device):
            _native.check(_native.lib().gs2m_texture_2d_clamp_backward(uv.numel() // 2, ch, W, H, g.data_ptr(), uv.data_ptr(),
                                                                       dy.data_ptr(), _stream(uv.device)), "gs2m_texture_2d_clamp_backward")
        return None, g


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if (uv_da is not None or mip_level_bias is not None) else "linear"
    if uv_da is not None:
        raise NotImplementedError("nvdiffrast(texture) on MI355X: uv_da (screen-space derivatives) is not implemented; the "
                                  "reference selects mip levels through mip_level_bias only (pbr/shade.py:173-179)")
    if max_mip_level is not None:
        raise NotImplementedError("nvdiffrast(texture) on MI355X: max_mip_level is not implemented")
    tex, uv = _f32(tex, "tex"), _f32(uv, "uv")
    if uv.requires_grad or (mip_level_bias is not None and mip_level_bias.requires_grad):
        raise NotImplementedError("nvdiffrast(texture) on MI355X: gradients with respect to uv / mip_level_bias are not "
                                  "implemented (the reference detaches both); detach them")
    if boundary_mode == "cube":
        if tex.dim() != 5 or tex.shape[0] != 1 or tex.shape[1] != 6 or tex.shape[2] != tex.shape[3] or not 1 <= tex.shape[4] <= 4:
            raise RuntimeError("nvdiffrast(texture): cube map must have shape (1, 6, w, w, C), C <= 4")
        if uv.dim() != 4 or uv.shape[-1] != 3:
            raise RuntimeError("nvdiffrast(texture): cube map lookups take uv of shape (N, h, w, 3)")
        if filter_mode == "linear":
            if mip is not None or mip_level_bias is not None:
                raise RuntimeError("nvdiffrast(texture): filter_mode='linear' takes no mip stack / bias")
            return _CubeTexture.apply(uv, None, tex)
        if filter_mode == "linear-mipmap-linear":
            if mip is None or mip_level_bias is None:
                raise NotImplementedError("nvdiffrast(texture) on MI355X: 'linear-mipmap-linear' needs an explicit `mip` list "
                                          "and `mip_level_bias` (internal mip construction is not implemented)")
            levels = [tex] + [_f32(m, "mip") for m in mip]
            for a, b in zip(levels[:-1], levels[1:]):
                if b.shape != (1, 6, a.shape[2] // 2, a.shape[3] // 2, a.shape[4]):
                    raise RuntimeError("nvdiffrast(texture): every mip level must halve the previous one")
            bias = _f32(mip_level_bias, "mip_level_bias")
            if bias.shape != uv.shape[:-1]:
                raise RuntimeError("nvdiffrast(texture): mip_level_bias must have shape (N, h, w)")
            return _CubeTexture.apply(uv, bias, *levels)
        raise NotImplementedError(f"nvdiffrast(texture) on MI355X: filter_mode={filter_mode!r} with boundary_mode='cube'")
    if boundary_mode == "clamp" and filter_mode == "linear":
        if tex.dim() != 4 or tex.shape[0] != 1 or not 1 <= tex.shape[3] <= 4 or uv.dim() != 4 or uv.shape[-1] != 2:
            raise RuntimeError("nvdiffrast(texture): 2-D lookups take tex (1, H, W, C), C <= 4, and uv (N, h, w, 2)")
        return _Texture2DClamp.apply(uv, tex)
    raise NotImplementedError(f"nvdiffrast(texture) on MI355X: filter_mode={filter_mode!r}, boundary_mode={boundary_mode!r} is not "
                              "implemented (cube/linear, cube/linear-mipmap-linear and clamp/linear are)")
