"""Learnable cube-map environment light with split-sum prefiltering (pbr/light.py:13-126 of the reference, same class,
method and attribute names), on this repository's HIP operators: `nvdiffrast.torch.texture` (csrc/texture.hip) and
`render_utils.diffuse_cubemap` / `specular_cubemap` (csrc/cubemap.hip)."""
from typing import List, Optional

import numpy as np
import nvdiffrast.torch as dr
import torch
import torch.nn as nn
import torch.nn.functional as F

from render_utils import diffuse_cubemap, specular_cubemap


def cube_to_dir(s: int, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Face s, face coordinates x, y in [-1, 1] -> (unnormalised) direction; the OpenGL cube-map convention."""
    one = torch.ones_like(x)
    table = ((one, -y, -x), (-one, -y, x), (x, one, y), (x, -one, -y), (x, -y, one), (-x, -y, -one))
    return torch.stack(table[s], dim=-1)


_dirs_cache = {}


def _texel_center_dirs(res, device):
    """(6, res, res, 3) unit directions of the texel centres (constant per resolution: cached; the reference rebuilds
    them in every backward call, ~25 small launches per level)."""
    key = (str(device), int(res))
    if key not in _dirs_cache:
        _dirs_cache[key] = _build_texel_center_dirs(res, device)
    return _dirs_cache[key]


def _build_texel_center_dirs(res, device):
    c = torch.linspace(-1.0 + 1.0 / res, 1.0 - 1.0 / res, res, device=device)
    gy, gx = torch.meshgrid(c, c, indexing="ij")
    return torch.stack([F.normalize(cube_to_dir(s, gx, gy), p=2, dim=-1) for s in range(6)], dim=0)


class cubemap_mip(torch.autograd.Function):
    """2x2 average pooling of every face.  As in the reference (:29-48) the backward is not the plain adjoint: the
    coarse gradient is looked up bilinearly -- across cube edges -- at the fine texel centres, times 0.25."""

    @staticmethod
    def forward(ctx, cubemap: torch.Tensor) -> torch.Tensor:
        y = F.avg_pool2d(cubemap.permute(0, 3, 1, 2), (2, 2))
        return y.permute(0, 2, 3, 1).contiguous()

    @staticmethod
    def backward(ctx, dout: torch.Tensor) -> torch.Tensor:
        res = dout.shape[1] * 2
        v = _texel_center_dirs(res, dout.device).view(1, 6 * res, res, 3)
        out = dr.texture((dout * 0.25)[None, ...].contiguous(), v.contiguous(), filter_mode="linear", boundary_mode="cube")
        return out.view(6, res, res, dout.shape[-1])


class CubemapLight(nn.Module):
    LIGHT_MIN_RES = 16
    MIN_ROUGHNESS = 0.04
    MAX_ROUGHNESS = 0.5

    def __init__(self, base_res=512, scale=0.5, bias=0.25, device="cuda") -> None:
        super().__init__()
        self.mtx = None
        self.base = nn.Parameter(torch.rand(6, base_res, base_res, 3, dtype=torch.float32, device=device) * scale + bias)
        self.register_parameter("env_base", self.base)

    def xfm(self, mtx) -> None:
        self.mtx = mtx

    def clamp_(self, min: Optional[float] = None, max: Optional[float] = None) -> None:
        self.base.data.clamp_(min, max)

    def get_mip(self, roughness: torch.Tensor) -> torch.Tensor:
        """Roughness -> fractional level of the specular stack: [MIN, MAX] spreads over levels 0 .. n-2, (MAX, 1] over the last step."""
        n = len(self.specular)
        lo = (torch.clamp(roughness, self.MIN_ROUGHNESS, self.MAX_ROUGHNESS) - self.MIN_ROUGHNESS) / (self.MAX_ROUGHNESS - self.MIN_ROUGHNESS) * (n - 2)
        hi = (torch.clamp(roughness, self.MAX_ROUGHNESS, 1.0) - self.MAX_ROUGHNESS) / (1.0 - self.MAX_ROUGHNESS) + n - 2
        return torch.where(roughness < self.MAX_ROUGHNESS, lo, hi)

    def build_mips(self, cutoff: float = 0.99) -> None:
        self.specular = [self.base]
        while self.specular[-1].shape[1] > self.LIGHT_MIN_RES:
            self.specular += [cubemap_mip.apply(self.specular[-1])]
        self.diffuse = diffuse_cubemap(self.specular[-1])
        n = len(self.specular)
        for idx in range(n - 1):
            roughness = (idx / (n - 2)) * (self.MAX_ROUGHNESS - self.MIN_ROUGHNESS) + self.MIN_ROUGHNESS
            self.specular[idx] = specular_cubemap(self.specular[idx], roughness, cutoff)
        self.specular[-1] = specular_cubemap(self.specular[-1], 1.0, cutoff)

    def export_envmap(self, filename: Optional[str] = None, res: List[int] = [512, 1024], return_img: bool = False) -> Optional[torch.Tensor]:
        """Lat-long image of the base cube map (:101-126).  `filename`: .npy, or an image format PIL can write."""
        gy, gx = torch.meshgrid(torch.linspace(0.0 + 1.0 / res[0], 1.0 - 1.0 / res[0], res[0], device=self.base.device),
                                torch.linspace(-1.0 + 1.0 / res[1], 1.0 - 1.0 / res[1], res[1], device=self.base.device), indexing="ij")
        sintheta, costheta = torch.sin(gy * np.pi), torch.cos(gy * np.pi)
        sinphi, cosphi = torch.sin(gx * np.pi), torch.cos(gx * np.pi)
        reflvec = torch.stack((sintheta * sinphi, costheta, -sintheta * cosphi), dim=-1)
        color = dr.texture(self.base.detach()[None, ...], reflvec[None, ...].contiguous(), filter_mode="linear", boundary_mode="cube")[0]
        if return_img:
            return color
        img = color.clamp(min=0.0).cpu().numpy()
        if filename.endswith(".npy"):
            np.save(filename, img)
        else:
            from PIL import Image
            Image.fromarray((np.clip(img, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)).save(filename)
