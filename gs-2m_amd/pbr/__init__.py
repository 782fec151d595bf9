"""Deferred PBR shading of the rasterizer's G-buffer on MI355X: same package, function names and
result keys as the reference's `pbr/` (pbr/__init__.py:1-56), on this repository's HIP operators."""
import torch
import torch.nn.functional as F

from .light import CubemapLight
from .shade import (aces_film, envBRDF_approx, get_brdf_lut, linear_to_srgb, pbr_shading, pbr_shading_fused, rgb_to_srgb, saturate_dot,
                    shading_inputs_fused, srgb_to_linear, srgb_to_rgb)

__all__ = ["CubemapLight", "get_brdf_lut", "pbr_shading", "pbr_shading_fused", "saturate_dot", "linear_to_srgb", "srgb_to_linear", "pbr_render"]


def pbr_render(scene, viewpoint_cam, canonical_rays, render_pkg, metallic, gamma=False, fused=True):
    """scene: anything with `.cubemap` (CubemapLight) and `.brdf_lut`.  Gradients reach the environment light, the albedo
    map and -- when `metallic` -- the metallic map; normals, roughness and the estimated metallic are detached, as in
    the reference (:25-43)."""
    scene.cubemap.build_mips()
    H, W = viewpoint_cam.image_height, viewpoint_cam.image_width
    # view directions: constant per camera, kept on it (an (H W, 3) x (3, 3) product is a BLAS call on ROCm, 0.13 ms at 1080p)
    cache = getattr(viewpoint_cam, "_pbr_view_dirs", None)
    if cache is None or cache[0] != (canonical_rays.data_ptr(), canonical_rays._version, H, W):
        c2w = viewpoint_cam.world_view_transform[:3, :3]
        view_dirs = F.normalize(-canonical_rays @ c2w.T, p=2, dim=-1).reshape(H, W, 3)
        try:
            viewpoint_cam._pbr_view_dirs = ((canonical_rays.data_ptr(), canonical_rays._version, H, W), view_dirs)
        except AttributeError:
            pass
    else:
        view_dirs = cache[1]

    if fused and render_pkg["normal_map"].is_cuda:  # the preparation below as one kernel each way, then the fused shading
        H_, W_ = render_pkg["normal_map"].shape[-2:]
        normals, albedo, rough, metal = shading_inputs_fused(render_pkg["normal_map"], render_pkg["albedo_map"], render_pkg["roughness_map"],
                                                             render_pkg["alpha_map"], render_pkg["metallic_map"] if metallic else None, 0.04, 1.0)
        pkg = pbr_shading_fused(scene.cubemap, normals, view_dirs, albedo, rough, metallic=metal, brdf_lut=scene.brdf_lut, gamma=gamma)
        pkg.update({"roughness_map": rough.reshape(1, H_, W_), "metallic_map": metal.reshape(1, H_, W_)})
        return pkg
    normal_map = render_pkg["normal_map"].detach()
    normal_map = torch.where(torch.norm(normal_map, dim=0, keepdim=True) > 0, F.normalize(normal_map, dim=0, p=2), normal_map)
    albedo_map = render_pkg["albedo_map"].clamp(0, 1)
    metallic_map = render_pkg["metallic_map"]
    roughness_map = render_pkg["roughness_map"]
    if not metallic:  # estimate it from the roughness
        metallic_map = (render_pkg["alpha_map"].detach() * (1.0 - roughness_map).clamp(0, 1)).detach()
    rmin, rmax = 0.04, 1.0
    roughness_map = (roughness_map * (rmax - rmin) + rmin).detach()

    if fused:  # the same shading as one kernel each way (include/gs2m_pbr.h); `fused=False`: the reference's op-by-op form
        pkg = pbr_shading_fused(scene.cubemap, normal_map.permute(1, 2, 0), view_dirs, albedo_map.permute(1, 2, 0),
                                roughness_map.permute(1, 2, 0), metallic=metallic_map.permute(1, 2, 0), brdf_lut=scene.brdf_lut, gamma=gamma)
        pkg.update({"roughness_map": roughness_map, "metallic_map": metallic_map})
        return pkg
    pkg = pbr_shading(light=scene.cubemap, normals=normal_map.permute(1, 2, 0), view_dirs=view_dirs, albedo=albedo_map.permute(1, 2, 0),
                      roughness=roughness_map.permute(1, 2, 0), metallic=metallic_map.permute(1, 2, 0),
                      occlusion=torch.ones_like(roughness_map).permute(1, 2, 0), irradiance=torch.zeros_like(roughness_map).permute(1, 2, 0),
                      brdf_lut=scene.brdf_lut, gamma=gamma)
    pkg.update({"roughness_map": roughness_map, "metallic_map": metallic_map})
    return pkg
