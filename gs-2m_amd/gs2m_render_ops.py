"""Autograd wrappers of the fused render() pre/post-processing kernels (csrc/render_ops.hip).

`pack_features` replaces pc.get_normals + the camera-space products + the feature-row packing of the reference's
render() (gaussian_renderer/__init__.py:83-96, scene/gaussian_model.py:146-160); `gbuffer_post` replaces its G-buffer
post-processing (:126-141).  Same values, same gradients as the PyTorch ops they stand for (tests/test_render_ops_gpu.py);
HIP tensors only -- there is no CPU path, the callers keep the reference's PyTorch formulation for that."""
import torch

import gs2m_arena as _arena
import gs2m_native as _native


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return _native.stream_ptr()


def _f32c(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"gs2m render ops: `{name}` must be on a HIP (cuda) device; there is no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"gs2m render ops: `{name}` must be float32, got {t.dtype}")
    return t.contiguous()


class _PackFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scales, rotations, albedo, roughness, metallic, campos, view, z_depth, blend_metallic):
        xyz, scales, rotations = _f32c(xyz, "xyz"), _f32c(scales, "scales"), _f32c(rotations, "rotations")
        albedo, roughness, metallic = _f32c(albedo, "albedo"), _f32c(roughness, "roughness"), _f32c(metallic, "metallic")
        campos, view = _f32c(campos, "camera_center"), _f32c(view, "world_view_transform")
        P = xyz.shape[0]
        features = torch.empty((P, 10), dtype=torch.float32, device=xyz.device)
        with _native.device_guard(xyz.device):
            _native.check(_native.lib().gs2m_pack_features_forward(
                P, _ptr(xyz), _ptr(scales), _ptr(rotations), _ptr(albedo), _ptr(roughness), _ptr(metallic), _ptr(campos),
                _ptr(view), int(bool(z_depth)), int(bool(blend_metallic)), _ptr(features), _stream()), "gs2m_pack_features_forward")
        ctx.save_for_backward(xyz, scales, rotations, campos, view)
        ctx.flags = (int(bool(z_depth)), int(bool(blend_metallic)))
        return features

    @staticmethod
    def backward(ctx, dF):
        xyz, scales, rotations, campos, view = ctx.saved_tensors
        dF = _f32c(dF, "grad_features")
        P = xyz.shape[0]
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=xyz.device)
        d_xyz, d_rot, d_alb, d_rgh, d_met = e(P, 3), e(P, 4), e(P, 3), e(P, 1), e(P, 1)
        with _native.device_guard(xyz.device):
            _native.check(_native.lib().gs2m_pack_features_backward(
                P, _ptr(xyz), _ptr(scales), _ptr(rotations), _ptr(campos), _ptr(view), ctx.flags[0], ctx.flags[1], _ptr(dF),
                _ptr(d_xyz), _ptr(d_rot), _ptr(d_alb), _ptr(d_rgh), _ptr(d_met), _stream()), "gs2m_pack_features_backward")
        # metallic is not read unless blend_metallic: no gradient then (None, as with the PyTorch formulation)
        return d_xyz, None, d_rot, d_alb, d_rgh, (d_met if ctx.flags[1] else None), None, None, None, None


def pack_features(xyz, scales, rotations, albedo, roughness, metallic, camera_center, world_view_transform,
                  z_depth=False, blend_metallic=False):
    """-> features (P, 10) = [1, distance, normal(3), albedo(3), roughness, metallic | 0] (GR:83-96, GM:146-160)."""
    return _PackFeatures.apply(xyz, scales, rotations, albedo, roughness, metallic, camera_center, world_view_transform,
                               z_depth, blend_metallic)


class _GBufferPost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buffer, rays, view, z_depth):
        buffer, view = _f32c(buffer, "buffer"), _f32c(view, "world_view_transform")
        rays = None if rays is None else _f32c(rays, "rays")
        _, H, W = buffer.shape
        mask = torch.empty((1, H, W), dtype=torch.bool, device=buffer.device)  # one byte per pixel, the kernel writes 0 / 1
        local_normal = torch.empty((3, H, W), dtype=torch.float32, device=buffer.device)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=buffer.device)
        with _native.device_guard(buffer.device):
            _native.check(_native.lib().gs2m_gbuffer_post_forward(
                W, H, _ptr(buffer), _ptr(rays), _ptr(view), int(bool(z_depth)), _ptr(mask), _ptr(local_normal), _ptr(depth),
                _stream()), "gs2m_gbuffer_post_forward")
        ctx.save_for_backward(buffer, rays, view)
        ctx.z_depth = int(bool(z_depth))
        ctx.mark_non_differentiable(mask)
        return mask, local_normal, depth

    @staticmethod
    def backward(ctx, _dmask, d_local_normal, d_depth):
        buffer, rays, view = ctx.saved_tensors
        _, H, W = buffer.shape
        d_buffer = torch.zeros_like(buffer)  # channels 0, 5..9 get no gradient from here
        dl = None if d_local_normal is None else _f32c(d_local_normal, "grad_local_normal_map")
        dd = None if d_depth is None else _f32c(d_depth, "grad_depth_map")
        with _native.device_guard(buffer.device):
            _native.check(_native.lib().gs2m_gbuffer_post_backward(
                W, H, _ptr(buffer), _ptr(rays), _ptr(view), ctx.z_depth, _ptr(dl), _ptr(dd), _ptr(d_buffer), _stream()),
                "gs2m_gbuffer_post_backward")
        return d_buffer, None, None, None


def gbuffer_post(buffer, rays, world_view_transform, z_depth=False):
    """buffer (10,H,W) -> (normal_mask (1,H,W) bool, local_normal_map (3,H,W), depth_map (1,H,W)) (GR:126-141)."""
    return _GBufferPost.apply(buffer, rays, world_view_transform, z_depth)


class _GBufferMaps(torch.autograd.Function):
    """gbuffer_post plus the maps render() hands out as channel slices of the buffer, as ONE autograd node: the slices
    are returned as views (no copy) and their gradients are folded into the single kernel that writes dL/dbuffer."""

    @staticmethod
    def forward(ctx, buffer, rays, view, z_depth):
        b = _f32c(buffer, "buffer")
        mask, local_normal, depth = _GBufferPost.forward(ctx, b, rays, view, z_depth)
        ctx.set_materialize_grads(False)  # maps the loss never touches arrive as None (the kernel takes NULL), not as zero-filled frames
        return b[0:1], b[1:2], b[2:5], b[5:8], b[8:9], b[9:10], mask, local_normal, depth

    @staticmethod
    def backward(ctx, d_alpha, d_dist, d_normal, d_albedo, d_rough, d_metal, _dmask, d_local_normal, d_depth):
        buffer, rays, view = ctx.saved_tensors
        _, H, W = buffer.shape
        d_buffer = torch.empty_like(buffer)  # the kernel writes all ten channels
        c = lambda t, n: None if t is None else _f32c(t, n)
        with _native.device_guard(buffer.device):
            _native.check(_native.lib().gs2m_gbuffer_maps_backward(
                W, H, _ptr(buffer), _ptr(rays), _ptr(view), ctx.z_depth, _ptr(c(d_local_normal, "grad_local_normal_map")),
                _ptr(c(d_depth, "grad_depth_map")), _ptr(c(d_alpha, "grad_alpha_map")), _ptr(c(d_dist, "grad_distance_map")),
                _ptr(c(d_normal, "grad_normal_map")), _ptr(c(d_albedo, "grad_albedo_map")), _ptr(c(d_rough, "grad_roughness_map")),
                _ptr(c(d_metal, "grad_metallic_map")), _ptr(d_buffer), _stream()), "gs2m_gbuffer_maps_backward")
        return d_buffer, None, None, None


def gbuffer_maps(buffer, rays, world_view_transform, z_depth=False):
    """buffer (10,H,W) -> (alpha_map, distance_map, normal_map, albedo_map, roughness_map, metallic_map  [channel slices],
    normal_mask, local_normal_map, depth_map): everything render() derives from the G-buffer (GR:126-163)."""
    return _GBufferMaps.apply(buffer, rays, world_view_transform, z_depth)


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, rotation, opacity, albedo, roughness, metallic):
        raw = [_f32c(t, n) for t, n in zip((scaling, rotation, opacity, albedo, roughness, metallic),
                                           ("_scaling", "_rotation", "_opacity", "_albedo", "_roughness", "_metallic"))]
        P = raw[0].shape[0]
        out = [torch.empty_like(t) for t in raw]
        with _native.device_guard(raw[0].device):
            _native.check(_native.lib().gs2m_activate_forward(P, *[_ptr(t) for t in raw], *[_ptr(t) for t in out], _stream()),
                          "gs2m_activate_forward")
        ctx.save_for_backward(raw[1], out[0], *out[2:])
        ctx.set_materialize_grads(False)  # an activation the loss never reaches gets no gradient (None), as with the getters
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        rotation, scales, opac, alb, rgh, met = ctx.saved_tensors
        P = rotation.shape[0]
        need = [g is not None and ctx.needs_input_grad[k] for k, g in enumerate(grads)]
        g = [_f32c(t, "grad") if n else None for t, n in zip(grads, need)]
        like = (scales, rotation, opac, alb, rgh, met)
        # the raw-parameter gradients side by side in one registered arena (gs2m_arena): data-parallel training sums them
        # in place with one collective (they become the parameters' .grad as they are)
        names = ("scaling", "rotation", "opacity", "albedo", "roughness", "metallic")
        arena = _arena.GradArena(rotation.device, [(nm, t.shape) for nm, t, n in zip(names, like, need) if n], key="activate")
        d = [arena[nm] if n else None for nm, n in zip(names, need)]
        with _native.device_guard(rotation.device):
            _native.check(_native.lib().gs2m_activate_backward(
                P, _ptr(rotation), _ptr(scales), _ptr(opac), _ptr(alb), _ptr(rgh), _ptr(met), *[_ptr(t) for t in g],
                *[_ptr(t) for t in d], _stream()), "gs2m_activate_backward")
        return tuple(d)


def activate(scaling, rotation, opacity, albedo, roughness, metallic):
    """raw parameters -> (scales, rotations, opacity, albedo, roughness, metallic) as the GaussianModel getters compute
    them (exp, F.normalize, sigmoid x4; scene/gaussian_model.py:113-144), one launch forward and one backward."""
    return _Activate.apply(scaling, rotation, opacity, albedo, roughness, metallic)


class _SobelNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, alpha, bg, view, fx, fy, cx, cy):
        depth, alpha, bg, view = _f32c(depth, "depth"), _f32c(alpha, "alpha_map"), _f32c(bg, "bg_color"), _f32c(view, "world_view_transform")
        H, W = depth.shape
        out = torch.empty((3, H, W), dtype=torch.float32, device=depth.device)
        with _native.device_guard(depth.device):
            _native.check(_native.lib().gs2m_sobel_normal_forward(W, H, _ptr(depth), _ptr(alpha), _ptr(bg), _ptr(view), fx, fy, cx, cy,
                                                                  _ptr(out), _stream()), "gs2m_sobel_normal_forward")
        ctx.save_for_backward(depth, alpha, bg, view)
        ctx.k = (fx, fy, cx, cy)
        return out

    @staticmethod
    def backward(ctx, g):
        depth, alpha, bg, view = ctx.saved_tensors
        g = _f32c(g, "grad_sobel_map")
        H, W = depth.shape
        d_depth, d_alpha = torch.empty_like(depth), torch.empty_like(alpha)
        with _native.device_guard(depth.device):
            _native.check(_native.lib().gs2m_sobel_normal_backward(W, H, _ptr(depth), _ptr(alpha), _ptr(bg), _ptr(view), *ctx.k,
                                                                   _ptr(g), _ptr(d_depth), _ptr(d_alpha), _stream()),
                          "gs2m_sobel_normal_backward")
        return d_depth, d_alpha, None, None, None, None, None, None


def sobel_normal(depth, alpha_map, bg_color, world_view_transform, fx, fy, cx, cy):
    """depth (H,W), alpha (H,W), bg (3) -> (3,H,W): render_normal_from_depth_map (GR:167-175, utils/normal_utils.py:3-72)
    for a zero-skew pinhole camera (intrinsic [[fx,0,cx],[0,fy,cy],[0,0,1]] as get_calib_matrix_nerf() builds it)."""
    return _SobelNormal.apply(depth, alpha_map, bg_color, world_view_transform, float(fx), float(fy), float(cx), float(cy))
