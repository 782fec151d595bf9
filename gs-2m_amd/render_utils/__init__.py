"""Drop-in for the two `render_utils` operators the reference's PBR stage imports (pbr/light.py:10:
`from render_utils import diffuse_cubemap, specular_cubemap`; submodules/render-utils/render_utils/ops.py:336-403) on
MI355X.  Same names and argument meaning; hand-written HIP behind include/gs2m_cubemap.h (csrc/cubemap.hip).  The
backward passes are deterministic gathers; no bounds table is built (the boxes are an acceleration structure in the
reference and do not change the result).  There is no CPU path."""
import ctypes as C

import numpy as np
import torch

import gs2m_native as _native


def _check(t, name, ch):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"render_utils: `{name}` must be a CUDA tensor (HIP kernel; there is no CPU path)")
    if t.dtype != torch.float32 or t.dim() != 4 or t.shape[0] != 6 or t.shape[1] != t.shape[2] or t.shape[3] != ch:
        raise RuntimeError(f"render_utils: `{name}` must be a float32 (6, res, res, {ch}) tensor, got {tuple(t.shape)} {t.dtype}")
    return t.contiguous()


def _stream(dev):
    return C.c_void_p(_native.stream_ptr(dev))


class _diffuse_cubemap_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cubemap):
        cubemap = _check(cubemap, "cubemap", 3)
        out = torch.empty_like(cubemap)
        with _native.device_guard(cubemap.device):
            _native.check(_native.lib().gs2m_diffuse_cubemap_forward(cubemap.shape[1], cubemap.data_ptr(), out.data_ptr(),
                                                                     _stream(cubemap.device)), "gs2m_diffuse_cubemap_forward")
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _check(dout, "grad", 3)
        g = torch.empty_like(dout)
        with _native.device_guard(dout.device):
            _native.check(_native.lib().gs2m_diffuse_cubemap_backward(dout.shape[1], dout.data_ptr(), g.data_ptr(),
                                                                      _stream(dout.device)), "gs2m_diffuse_cubemap_backward")
        return g


def diffuse_cubemap(cubemap, use_python=False):
    assert not use_python
    return _diffuse_cubemap_func.apply(cubemap)


_tables = {}


def _texel_table(res, device):
    """(res,) separable texel-area factors of one cube-map level, cached per (device, res)."""
    key = (str(device), int(res))
    if key not in _tables:
        t = torch.empty((res,), dtype=torch.float32, device=device)
        with _native.device_guard(device):
            _native.check(_native.lib().gs2m_cubemap_texel_table(res, t.data_ptr(), _stream(device)), "gs2m_cubemap_texel_table")
        _tables[key] = t
    return _tables[key]


class _specular_cubemap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cubemap, roughness, costheta_cutoff):
        cubemap = _check(cubemap, "cubemap", 3)
        res = cubemap.shape[1]
        out = torch.empty((6, res, res, 4), dtype=torch.float32, device=cubemap.device)
        with _native.device_guard(cubemap.device):
            _native.check(_native.lib().gs2m_specular_cubemap_forward(res, float(roughness), float(costheta_cutoff), _texel_table(res, cubemap.device).data_ptr(),
                                                                      cubemap.data_ptr(), out.data_ptr(), _stream(cubemap.device)), "gs2m_specular_cubemap_forward")
        ctx.args = (float(roughness), float(costheta_cutoff))
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _check(dout, "grad", 4)
        res = dout.shape[1]
        g = torch.empty((6, res, res, 3), dtype=torch.float32, device=dout.device)
        with _native.device_guard(dout.device):
            _native.check(_native.lib().gs2m_specular_cubemap_backward(res, ctx.args[0], ctx.args[1], _texel_table(res, dout.device).data_ptr(), dout.data_ptr(), g.data_ptr(),
                                                                       _stream(dout.device)), "gs2m_specular_cubemap_backward")
        return g, None, None


class _specular_cubemap_normalized(torch.autograd.Function):
    """specular_cubemap's `out[..., 0:3] / out[..., 3:]` folded into the operator (forward: gather + divide; backward: prescale +
    gather) -- two launches each way instead of ~16."""

    @staticmethod
    def forward(ctx, cubemap, roughness, costheta_cutoff):
        cubemap = _check(cubemap, "cubemap", 3)
        res = cubemap.shape[1]
        raw = torch.empty((6, res, res, 4), dtype=torch.float32, device=cubemap.device)
        out = torch.empty_like(cubemap)
        with _native.device_guard(cubemap.device):
            _native.check(_native.lib().gs2m_specular_cubemap_normalized_forward(
                res, float(roughness), float(costheta_cutoff), _texel_table(res, cubemap.device).data_ptr(), cubemap.data_ptr(), raw.data_ptr(),
                out.data_ptr(), _stream(cubemap.device)), "gs2m_specular_cubemap_normalized_forward")
        ctx.save_for_backward(raw)
        ctx.args = (float(roughness), float(costheta_cutoff))
        return out

    @staticmethod
    def backward(ctx, dout):
        (raw,) = ctx.saved_tensors
        dout = _check(dout, "grad", 3)
        res = dout.shape[1]
        scratch = torch.empty_like(raw)
        g = torch.empty_like(dout)
        with _native.device_guard(dout.device):
            _native.check(_native.lib().gs2m_specular_cubemap_normalized_backward(
                res, ctx.args[0], ctx.args[1], _texel_table(res, dout.device).data_ptr(), raw.data_ptr(), dout.data_ptr(), scratch.data_ptr(),
                g.data_ptr(), _stream(dout.device)), "gs2m_specular_cubemap_normalized_backward")
        return g, None, None


_cutoff_cache = {}


def ndf_cutoff(roughness, cutoff):
    """cos of the angle inside which `cutoff` of the GGX NDF (alpha = roughness^2), summed over a million equally spaced
    angles in [0, pi/2], lies (render_utils/ops.py:373-388 without the bounds table)."""
    key = (float(roughness), float(cutoff))
    if key not in _cutoff_cache:
        costheta = np.cos(np.linspace(0, np.pi / 2.0, 1000000))
        a2 = roughness ** 4
        c = np.clip(costheta, 0.0, 1.0)
        d = (c * a2 - c) * c + 1.0
        D = np.cumsum(a2 / (d * d * np.pi))
        _cutoff_cache[key] = float(costheta[np.argmax(D >= D[-1] * cutoff)])
    return _cutoff_cache[key]


def specular_cubemap(cubemap, roughness, cutoff=0.99, use_python=False):
    assert not use_python
    assert cubemap.shape[0] == 6 and cubemap.shape[1] == cubemap.shape[2], "Bad shape for cubemap tensor: %s" % str(cubemap.shape)
    return _specular_cubemap_normalized.apply(cubemap, roughness, ndf_cutoff(roughness, cutoff))
