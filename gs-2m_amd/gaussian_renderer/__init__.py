"""`render()` -- the caller of the rasterizer hot path, reproducing the reference's
gaussian_renderer/__init__.py:21-165 call for call on PyTorch-ROCm: same arguments, same
feature-row packing and feature_count rule (:86-96), same settings construction (:98-110),
same post-processing (:126-141) and the same 14(+1)-entry output dict (:143-163).

`pc` is any object with the reference GaussianModel's accessors (scene/gaussian_model.py:113-172):
get_xyz, get_opacity, get_albedo, get_roughness, get_metallic, get_scaling, get_rotation,
get_features, get_normals(camera_center), get_covariance(), active_sh_degree, max_sh_degree.
`viewpoint_camera` needs FoVx, FoVy, image_height, image_width, world_view_transform,
full_proj_transform, camera_center, get_rays(), get_calib_matrix_nerf() (scene/cameras.py).
gs2m_scene.py provides minimal implementations of both.
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
import gs2m_render_ops
from gs2m_scene import eval_sh, normal_from_depth_image


_zero_points_cache = {}


def _zero_points(P, dtype, device):
    key = (P, dtype, device)
    z = _zero_points_cache.get(key)
    if z is None:
        _zero_points_cache.clear()  # one size at a time (densification changes P)
        z = _zero_points_cache[key] = torch.zeros((P, 4), dtype=dtype, device=device)
    return z


_zero_colors_cache = {}


def _zero_colors(P, device):
    key = (P, device)
    z = _zero_colors_cache.get(key)
    if z is None:
        _zero_colors_cache.clear()
        z = _zero_colors_cache[key] = torch.zeros((P, 3), dtype=torch.float32, device=device)
    return z


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, geometry_stage=False, material_stage=False,
           sobel_normal=False, blend_metallic=False, shade=True):
    """Render the scene.  Background tensor (bg_color) must be on the GPU.
    `shade=False` (this repository's extension; the reference has no such argument): the caller wants the G-buffer only -- the
    multi-view term's neighbour view, whose depth and normal maps alone enter the loss (utils/loss_utils.py:253-276).  The colours go in
    as precomputed zeros instead of SH coefficients: `render` comes out black, and the forward does not read the (P,16,3) coefficients
    nor the backward write their gradient."""
    device = pc.get_xyz.device
    # zero tensor whose gradient carries the 2D (screen-space) mean gradients: the first two columns as
    # in 3DGS, the last two accumulate absolute values (GR:38-43)
    if device.type == "cuda" and bool(getattr(pipe, "fused_render_ops", True)):
        # A fresh LEAF over a cached block of zeros: the rasterizer never reads the values (only the gradient matters), so no
        # fill, no `+ 0`, and after backward `.grad` IS the rasterizer's dL/dmeans2D buffer (autograd takes it over; the
        # reference's non-leaf + retain_grad idiom costs a fill, an add and two (P,4) copies per view)
        screenspace_points = _zero_points(pc.get_xyz.shape[0], pc.get_xyz.dtype, device).detach().requires_grad_(True)
    else:
        screenspace_points = torch.zeros((pc.get_xyz.shape[0], 4), dtype=pc.get_xyz.dtype, requires_grad=True,
                                         device=device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)

    means3D = pc.get_xyz
    means2D = screenspace_points
    scales = None
    rotations = None
    cov3D_precomp = None
    # pipe.fused_render_ops: the six parameter activations (the model's getters: exp, normalize, sigmoid x4) as one
    # launch each way, when the model exposes the reference's raw parameter names
    raw = [getattr(pc, n, None) for n in ("_scaling", "_rotation", "_opacity", "_albedo", "_roughness", "_metallic")]
    if (bool(getattr(pipe, "fused_render_ops", True)) and bool(getattr(pipe, "fused_activations", True)) and not pipe.compute_cov3D_python
            and all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in raw)):
        scales, rotations, opacity, albedo, roughness, metallic = gs2m_render_ops.activate(*raw)
    else:
        opacity = pc.get_opacity
        albedo = pc.get_albedo
        roughness = pc.get_roughness
        metallic = pc.get_metallic
        if pipe.compute_cov3D_python:
            cov3D_precomp = pc.get_covariance()
        else:
            scales = pc.get_scaling
            rotations = pc.get_rotation

    shs = None
    shs_rest = None
    colors_precomp = None
    if not shade and device.type == "cuda":
        colors_precomp = _zero_colors(pc.get_xyz.shape[0], device)
    elif pipe.convert_SHs_python:
        shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dir_pp = (pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1))
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
        colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
    else:
        # the reference model keeps the SH coefficients as two parameters and concatenates them for every view
        # (get_features: 192 B per Gaussian, and the split again in the backward); this rasterizer takes the two
        # parts directly (GaussianRasterizer.forward(..., shs_rest=...), degree-3 layout only)
        dc, rest = getattr(pc, "_features_dc", None), getattr(pc, "_features_rest", None)
        if (bool(getattr(pipe, "split_sh", True)) and torch.is_tensor(dc) and torch.is_tensor(rest) and dc.is_cuda
                and dc.dim() == 3 and rest.dim() == 3 and dc.shape[1] == 1 and rest.shape[1] == 15):
            shs, shs_rest = dc, rest
        else:
            shs = pc.get_features

    feature_count = 9 if material_stage else 5 if geometry_stage else 1
    if blend_metallic:
        feature_count += 1
    # pipe.fused_render_ops (default on): the normals, the camera-space products, the feature packing and the
    # G-buffer post-processing below run as fused HIP kernels (gs2m_render_ops) -- same values and gradients as
    # the PyTorch formulation of the reference, which stays here as the alternative (and is what the fused ops
    # are tested against).  On ROCm the K = 3 matmuls of that formulation cost ten times the rasterizer.
    fused = bool(getattr(pipe, "fused_render_ops", True)) and means3D.is_cuda
    if fused:
        features = gs2m_render_ops.pack_features(
            means3D, scales if scales is not None else pc.get_scaling,
            rotations if rotations is not None else pc.get_rotation, albedo, roughness, metallic, viewpoint_camera.camera_center,
            viewpoint_camera.world_view_transform, z_depth=pipe.z_depth, blend_metallic=blend_metallic)
    else:
        normals = pc.get_normals(viewpoint_camera.camera_center)  # world space
        cam_normals = normals @ viewpoint_camera.world_view_transform[:3, :3]
        cam_points = means3D @ viewpoint_camera.world_view_transform[:3, :3] + viewpoint_camera.world_view_transform[3, :3]
        features = torch.zeros((means3D.shape[0], 10), dtype=torch.float32, device=device)
        features[:, 0] = 1.0  # alpha
        features[:, 1] = cam_points[:, 2] if pipe.z_depth else (cam_normals * cam_points).sum(dim=-1).abs()  # distance
        features[:, 2:5] = normals
        features[:, 5:8] = albedo
        features[:, 8:9] = roughness
        if blend_metallic:
            features[:, 9:10] = metallic

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=1.0,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        feature_count=feature_count)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    rendered_image, radii, observe, buffer = rasterizer(
        means3D=means3D, means2D=means2D, opacities=opacity, shs=shs, colors_precomp=colors_precomp,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, features=features, **({} if shs_rest is None else {"shs_rest": shs_rest}))

    normal_map = buffer[2:5, ...]  # (3, H, W)
    H, W = viewpoint_camera.image_height, viewpoint_camera.image_width
    distance_map = None if pipe.z_depth else buffer[1:2, ...]
    if fused:
        rays = None if pipe.z_depth else viewpoint_camera.get_rays().view(-1, 3)
        (alpha_map, dist_ch, normal_map, albedo_map, roughness_map, metallic_map, normal_mask, local_normal_map,
         depth_map) = gs2m_render_ops.gbuffer_maps(buffer, rays, viewpoint_camera.world_view_transform, z_depth=pipe.z_depth)
        distance_map = None if pipe.z_depth else dist_ch
    else:
        alpha_map, albedo_map, roughness_map, metallic_map = buffer[0:1, ...], buffer[5:8, ...], buffer[8:9, ...], buffer[9:10, ...]
        normal_mask = (normal_map != 0).all(0, keepdim=True).detach()
        local_normals = normal_map.permute(1, 2, 0).view(-1, 3)  # (H*W, 3)
        local_normals = local_normals @ viewpoint_camera.world_view_transform[:3, :3]
        local_normal_map = local_normals.reshape(H, W, 3).permute(2, 0, 1)
        depth_map = buffer[1:2, ...]
        if not pipe.z_depth:
            rays = viewpoint_camera.get_rays().view(-1, 3)
            denoms = torch.sum(local_normals * rays, dim=-1).view(1, H, W)
            depth_map = distance_map / -(denoms + 1e-8)

    out = {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
        "observe": observe,
        "alpha_map": alpha_map,
        "distance_map": distance_map if not pipe.z_depth else None,
        "depth_map": depth_map,
        "normal_map": normal_map,
        "albedo_map": albedo_map,
        "roughness_map": roughness_map,
        "metallic_map": metallic_map,
        "normal_mask": normal_mask,
        "local_normal_map": local_normal_map,
    }
    if sobel_normal:
        depth = out["depth_map"].squeeze(0)
        out["sobel_map"] = render_normal_from_depth_map(viewpoint_camera, depth, bg_color, out["alpha_map"][0], fused=fused)
    return out


def render_normal_from_depth_map(viewpoint_cam, depth, bg_color, alpha_map, fused=True):
    """depth (H,W), bg_color (3), alpha (H,W) -> (3,H,W); GR:167-175."""
    intrinsic, extrinsic = viewpoint_cam.get_calib_matrix_nerf()
    if fused and depth.is_cuda and not intrinsic.is_cuda:
        K = intrinsic.tolist()  # host tensor in the reference (scene/cameras.py:83-90): no device sync
        if K[0][1] == 0.0 and K[1][0] == 0.0 and K[2] == [0.0, 0.0, 1.0]:
            return gs2m_render_ops.sobel_normal(depth, alpha_map, bg_color, viewpoint_cam.world_view_transform,
                                                K[0][0], K[1][1], K[0][2], K[1][2])
    normal_ref = normal_from_depth_image(depth, intrinsic.to(depth.device), extrinsic.to(depth.device), view_space=False)
    background = bg_color[None, None, ...]
    normal_ref = normal_ref * alpha_map[..., None] + background * (1. - alpha_map[..., None])
    return normal_ref.permute(2, 0, 1)
