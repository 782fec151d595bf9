"""Multi-view geometric + photometric consistency loss: `multi_view_loss` of
utils/loss_utils.py:245-349 and what it needs of the reference's Scene (neighbour-camera tables, grey images, the pixel
grid: scene/__init__.py:125-141, 150-204).

Host-side Python as in the reference.  The photometric core -- plane-induced homography per sampled pixel, 7x7 patch
warp, two bilinear image lookups per patch point, NCC, and the backward to the rendered plane normals / distances --
is one fused HIP kernel each way (`patch_ncc`, include/gs2m_mvs.h); `patch_ncc_torch` is the same computation op by
op as the reference writes it (batched 3x3 matmuls, einsum, grid_sample, "ones" conv2d) and is what the fused kernel
is tested against.  The geometric part (reprojection error, normal agreement) stays PyTorch here.
"""
import ctypes as C
import math
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

import gs2m_native as _native


class MultiViewParams:
    """arguments/__init__.py:104-130 defaults read by this module."""
    multi_view_num = 8
    multi_view_ncc_weight = 0.15
    multi_view_geo_weight = 2e-3
    multi_view_ncc_scale = -1.0
    multi_view_max_angle = 30
    multi_view_min_dist = 0.01
    multi_view_max_dist = 1.5
    multi_view_sample_num = 102400
    multi_view_patch_size = 3
    mv_angle_threshold = 30
    mv_angle_factor = 2.0
    mv_occlusion_threshold = 5e-4
    mv_geo_weight_decay = 3.0
    reflection_threshold = 1.0
    nearby_cam_num = 16
    nearby_cam_max_angle = 60
    nearby_cam_min_angle = 10
    nearby_cam_min_dist = 0.05
    nearby_cam_max_dist = 2.5


# ---------------------------------------------------------------- the fused photometric core
class _PatchNCC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pixels, normals, dists, ref_gray, near_gray, M, b, Kinv, ncc_scale, patch):
        f = lambda t: t.contiguous().float()
        pixels, normals, dists, ref_gray, near_gray = f(pixels), f(normals), f(dists), f(ref_gray), f(near_gray)
        if not pixels.is_cuda:
            raise RuntimeError("patch_ncc: HIP kernel, there is no CPU path")
        h, w = ref_gray.shape[-2:]
        assert near_gray.shape[-2:] == (h, w), "both grey images must have the same size"
        N = pixels.shape[0]
        ncc = torch.empty((N, 1), dtype=torch.float32, device=pixels.device)
        consts = tuple((C.c_float * len(v))(*[float(x) for x in v]) for v in (M.reshape(-1).tolist(), b.reshape(-1).tolist(), Kinv.reshape(-1).tolist()))
        ctx.args = (N, w, h, consts, float(ncc_scale), int(patch))
        with _native.device_guard(pixels.device):
            _native.check(_native.lib().gs2m_patch_ncc_forward(
                N, pixels.data_ptr(), normals.data_ptr(), dists.data_ptr(), ref_gray.data_ptr(), near_gray.data_ptr(), w, h, *consts,
                float(ncc_scale), int(patch), ncc.data_ptr(), C.c_void_p(_native.stream_ptr(pixels.device))), "gs2m_patch_ncc_forward")
        ctx.save_for_backward(pixels, normals, dists, ref_gray, near_gray)
        return ncc

    @staticmethod
    def backward(ctx, d_ncc):
        pixels, normals, dists, ref_gray, near_gray = ctx.saved_tensors
        N, w, h, consts, ncc_scale, patch = ctx.args
        d_ncc = d_ncc.contiguous().float()
        dn, dd = torch.empty_like(normals), torch.empty_like(dists)
        with _native.device_guard(pixels.device):
            _native.check(_native.lib().gs2m_patch_ncc_backward(
                N, pixels.data_ptr(), normals.data_ptr(), dists.data_ptr(), ref_gray.data_ptr(), near_gray.data_ptr(), w, h, *consts,
                ncc_scale, patch, d_ncc.data_ptr(), dn.data_ptr(), dd.data_ptr(),
                C.c_void_p(_native.stream_ptr(pixels.device))), "gs2m_patch_ncc_backward")
        return None, dn, dd, None, None, None, None, None, None, None


class _GridSampleBorder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, grid):
        image, grid = image.contiguous().float(), grid.contiguous().float()
        if not image.is_cuda:
            raise RuntimeError("grid_sample_border: HIP kernel, there is no CPU path")
        Cc, H, W = image.shape
        N = grid.shape[0]
        out = torch.empty((N, Cc), dtype=torch.float32, device=image.device)
        with _native.device_guard(image.device):
            _native.check(_native.lib().gs2m_grid_sample_border_forward(N, Cc, H, W, image.data_ptr(), grid.data_ptr(), out.data_ptr(),
                                                                        C.c_void_p(_native.stream_ptr(image.device))),
                          "gs2m_grid_sample_border_forward")
        ctx.save_for_backward(image, grid)
        return out

    @staticmethod
    def backward(ctx, d_out):
        image, grid = ctx.saved_tensors
        Cc, H, W = image.shape
        d_img = torch.zeros_like(image) if ctx.needs_input_grad[0] else None
        d_grid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        d_out = d_out.contiguous().float()
        with _native.device_guard(image.device):
            _native.check(_native.lib().gs2m_grid_sample_border_backward(
                grid.shape[0], Cc, H, W, image.data_ptr(), grid.data_ptr(), d_out.data_ptr(), None if d_img is None else d_img.data_ptr(),
                None if d_grid is None else d_grid.data_ptr(), C.c_void_p(_native.stream_ptr(image.device))),
                "gs2m_grid_sample_border_backward")
        return d_img, d_grid


def set_deterministic(on=True):
    """The scatters of grid_sample_border's and mv_geo's backwards in 64-bit fixed point (integer atomics: bitwise reproducible, the
    default) or with fp32 atomics (include/gs2m_mvs.h)."""
    _native.lib().gs2m_mvs_set_deterministic(1 if on else 0)


def is_deterministic():
    return bool(_native.lib().gs2m_mvs_get_deterministic())


def grid_sample_border(image, grid):
    """F.grid_sample(image[None], grid.view(1, -1, 1, 2), mode='bilinear', padding_mode='border', align_corners=True) for a
    (C, H, W) image, C <= 4, returned as (N, C); gradients to both."""
    return _GridSampleBorder.apply(image, grid)


def _relative_pose(ref_cam, near_cam):
    """Y = A P + b takes a point from the reference camera's space to the neighbour's, Z = A2 Y + b2 back (host, float64; kept on
    the reference camera per neighbour)."""
    cache = ref_cam.__dict__.setdefault("_mvs_pose", {})
    if id(near_cam) not in cache:
        Vr, Vn = ref_cam.world_view_transform.double().cpu(), near_cam.world_view_transform.double().cpu()
        Rr, tr, Rn, tn = Vr[:3, :3].T, Vr[3, :3], Vn[:3, :3].T, Vn[3, :3]     # x_cam = R x_world + t
        A, A2 = Rn @ Rr.T, Rr @ Rn.T
        arr = lambda t: (C.c_float * t.numel())(*[float(x) for x in t.reshape(-1).tolist()])
        intr = lambda c: (C.c_float * 4)(float(c.Fx), float(c.Fy), float(c.Cx), float(c.Cy))
        cache[id(near_cam)] = (arr(A), arr(tn - A @ tr), arr(A2), arr(tr - A2 @ tn), intr(ref_cam), intr(near_cam))
    return cache[id(near_cam)]


class _MVGeo(torch.autograd.Function):
    """pixel_noise, angle, valid of multi_view_loss's geometric part as one kernel each way (include/gs2m_mvs.h)."""

    @staticmethod
    def forward(ctx, depth, normal, depth_n, normal_n, ref_cam, near_cam, occlusion):
        f = lambda t: t.contiguous().float()
        depth, normal, depth_n, normal_n = f(depth), f(normal), f(depth_n), f(normal_n)
        if not depth.is_cuda:
            raise RuntimeError("mv_geo: HIP kernel, there is no CPU path")
        H, W = depth.shape[-2:]
        Hn, Wn = depth_n.shape[-2:]
        consts = _relative_pose(ref_cam, near_cam)
        noise = torch.empty(H * W, dtype=torch.float32, device=depth.device)
        angle = torch.empty_like(noise)
        valid = torch.empty(H * W, dtype=torch.uint8, device=depth.device)
        with _native.device_guard(depth.device):
            _native.check(_native.lib().gs2m_mv_geo_forward(
                W, H, Wn, Hn, depth.data_ptr(), normal.data_ptr(), depth_n.data_ptr(), normal_n.data_ptr(), *consts, float(occlusion),
                noise.data_ptr(), angle.data_ptr(), valid.data_ptr(), C.c_void_p(_native.stream_ptr(depth.device))),
                "gs2m_mv_geo_forward")
        ctx.save_for_backward(depth, normal, depth_n, normal_n)
        ctx.args = (W, H, Wn, Hn, consts, float(occlusion))
        valid = valid.bool()
        ctx.mark_non_differentiable(valid)
        return noise, angle, valid

    @staticmethod
    def backward(ctx, d_noise, d_angle, _dv):
        depth, normal, depth_n, normal_n = ctx.saved_tensors
        W, H, Wn, Hn, consts, occlusion = ctx.args
        z = lambda t: torch.zeros(W * H, dtype=torch.float32, device=depth.device) if t is None else t.contiguous().float()
        d_noise, d_angle = z(d_noise), z(d_angle)
        dd, dn = torch.empty_like(depth), torch.empty_like(normal)
        ddn, dnn = torch.zeros_like(depth_n), torch.zeros_like(normal_n)
        with _native.device_guard(depth.device):
            _native.check(_native.lib().gs2m_mv_geo_backward(
                W, H, Wn, Hn, depth.data_ptr(), normal.data_ptr(), depth_n.data_ptr(), normal_n.data_ptr(), *consts, occlusion,
                d_noise.data_ptr(), d_angle.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddn.data_ptr(), dnn.data_ptr(),
                C.c_void_p(_native.stream_ptr(depth.device))), "gs2m_mv_geo_backward")
        return dd, dn, ddn, dnn, None, None, None


def mv_geo(depth, normal, depth_n, normal_n, ref_cam, near_cam, occlusion):
    """-> pixel_noise (H W), angle (H W) [rad], valid (H W) bool; gradients to the four maps."""
    return _MVGeo.apply(depth, normal, depth_n, normal_n, ref_cam, near_cam, occlusion)


class _MVGeoLoss(torch.autograd.Function):
    """utils/loss_utils.py:277-291 from mv_geo's per-pixel outputs as one launch each way (include/gs2m_loss.h:
    gs2m_mv_geo_loss_*): -> (weight * geo_loss, pixel_valid (bool), w_ncc = exp(-noise) on pixel_valid)."""

    @staticmethod
    def forward(ctx, noise, angle, valid, angle_threshold, decay, factor, weight):
        import gs2m_losses
        noise, angle, valid = noise.contiguous().float(), angle.contiguous().float(), valid.contiguous()
        n = noise.numel()
        dev = noise.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        pixel_valid = torch.empty(noise.shape, dtype=torch.bool, device=dev)
        w_ncc = torch.empty_like(noise)
        args = (float(angle_threshold), float(decay), float(factor), float(weight))
        with _native.device_guard(dev):
            _native.check(_native.lib().gs2m_mv_geo_loss_forward(
                n, noise.data_ptr(), angle.data_ptr(), valid.data_ptr(), *args, out.data_ptr(), pixel_valid.data_ptr(), w_ncc.data_ptr(),
                gs2m_losses._workspace(dev).data_ptr(), C.c_void_p(_native.stream_ptr(dev))), "gs2m_mv_geo_loss_forward")
        ctx.save_for_backward(noise, angle, valid, out)
        ctx.args = args
        ctx.mark_non_differentiable(pixel_valid, w_ncc)
        return out[0], pixel_valid, w_ncc

    @staticmethod
    def backward(ctx, g, _gv, _gw):
        noise, angle, valid, out = ctx.saved_tensors
        d_noise, d_angle = torch.empty_like(noise), torch.empty_like(angle)
        with _native.device_guard(noise.device):
            _native.check(_native.lib().gs2m_mv_geo_loss_backward(
                noise.numel(), noise.data_ptr(), angle.data_ptr(), valid.data_ptr(), *ctx.args, out.data_ptr(), g.contiguous().data_ptr(),
                d_noise.data_ptr(), d_angle.data_ptr(), C.c_void_p(_native.stream_ptr(noise.device))), "gs2m_mv_geo_loss_backward")
        return d_noise, d_angle, None, None, None, None, None


def mv_geo_loss(pixel_noise, angle, valid, opt):
    """-> (multi_view_geo_weight * geo_loss, pixel_valid, w_ncc): the fused form of the masked means below."""
    return _MVGeoLoss.apply(pixel_noise, angle, valid, opt.mv_angle_threshold * math.pi / 180.0, opt.mv_geo_weight_decay, opt.mv_angle_factor,
                            opt.multi_view_geo_weight)


def mv_geo_torch(depth, normal, depth_n, normal_n, ref_cam, near_cam, occlusion, pixels):
    """The same quantities op by op (utils/loss_utils.py:256-276)."""
    pts = _get_points_from_depth(ref_cam, depth)
    pts_near = _mm3(pts, near_cam.world_view_transform[:3, :3]) + near_cam.world_view_transform[3, :3]
    map_z, map_n, valid = _sample_depth_normal(pts_near, near_cam, {"depth_map": depth_n, "normal_map": normal_n}, fused=False)
    valid = valid & (pts_near[:, 2] - map_z <= occlusion)
    reproj = _reproject_points(near_cam, ref_cam, pts_near, map_z)
    noise = torch.norm(reproj - pixels.reshape(*reproj.shape), dim=-1)
    normals = _sample_normal_map(pixels, normal, fused=False)
    normals = normals / (normals.norm(dim=1, keepdim=True) + 1e-8)
    angle = torch.acos(torch.sum(normals * map_n, dim=1).clamp(-1 + 1e-6, 1 - 1e-6))
    return noise, angle, valid


def _homography_constants(ref_cam, near_cam, ncc_scale):
    """M = K_near R_rn K_ref^-1, b = K_near t_rn, K_ref^-1 (utils/loss_utils.py:319-327), on the host in float64; constant
    per camera pair, kept on the reference camera (reading the poses back is a device synchronisation)."""
    cache = ref_cam.__dict__.setdefault("_mvs_constants", {})
    key = (id(near_cam), float(ncc_scale))
    if key not in cache:
        cache[key] = _compute_homography_constants(ref_cam, near_cam, ncc_scale)
    return cache[key]


def _compute_homography_constants(ref_cam, near_cam, ncc_scale):
    Vr, Vn = ref_cam.world_view_transform.double().cpu(), near_cam.world_view_transform.double().cpu()
    rn_R = Vn[:3, :3].transpose(-1, -2) @ Vr[:3, :3]
    rn_t = -rn_R @ Vr[3, :3] + Vn[3, :3]
    Kn, Kinv = near_cam.get_K(ncc_scale).double().cpu(), ref_cam.get_inv_K(ncc_scale).double().cpu()
    return Kn @ rn_R @ Kinv, Kn @ rn_t, Kinv


def patch_ncc(pixels, normals, dists, ref_cam, near_cam, ncc_scale, patch):
    """pixels (N,2) full-resolution pixel coordinates, normals (N,3) camera-space plane normals, dists (N,) plane distances
    -> ncc (N,1) in [0, 2] (0 = perfectly correlated), mask (N,1) = ncc < 0.9.  Gradients to normals and dists."""
    M, b, Kinv = _homography_constants(ref_cam, near_cam, ncc_scale)
    ncc = _PatchNCC.apply(pixels, normals, dists, ref_cam.gray_image, near_cam.gray_image, M, b, Kinv, ncc_scale, patch)
    return ncc, ncc < 0.9


def patch_ncc_roughness(pixels, normals, dists, ref_cam, near_cam, ncc_scale, patch):
    """-> (ncc_gray, ncc_grad, std_mask), each (N, 1): the grey-value NCC, the NCC of the Sobel gradient magnitudes of the two
    patches, and the low-texture switch sqrt(ref_var) < 0.01 (utils/loss_utils.py:200-211).  No gradients (the reference
    evaluates this under no_grad)."""
    M, b, Kinv = _homography_constants(ref_cam, near_cam, ncc_scale)
    f = lambda t: t.detach().contiguous().float()
    pixels, normals, dists, rg, ng = f(pixels), f(normals), f(dists), f(ref_cam.gray_image), f(near_cam.gray_image)
    if not pixels.is_cuda:
        raise RuntimeError("patch_ncc_roughness: HIP kernel, there is no CPU path")
    h, w = rg.shape[-2:]
    N = pixels.shape[0]
    out = torch.empty((3, N, 1), dtype=torch.float32, device=pixels.device)
    consts = tuple((C.c_float * len(v))(*[float(x) for x in v]) for v in (M.reshape(-1).tolist(), b.reshape(-1).tolist(), Kinv.reshape(-1).tolist()))
    with _native.device_guard(pixels.device):
        _native.check(_native.lib().gs2m_patch_ncc_roughness(
            N, pixels.data_ptr(), normals.data_ptr(), dists.data_ptr(), rg.data_ptr(), ng.data_ptr(), w, h, *consts, float(ncc_scale), int(patch),
            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), C.c_void_p(_native.stream_ptr(pixels.device))),
            "gs2m_patch_ncc_roughness")
    return out[0], out[1], torch.sqrt(out[2]) < 0.01


# ---------------------------------------------------------------- the same, op by op (utils/loss_utils.py:303-349, 451-509)
def _patch_offsets(h_patch_size, device):
    o = torch.arange(-h_patch_size, h_patch_size + 1, device=device)
    return torch.stack(torch.meshgrid(o, o, indexing="xy")[::-1], dim=-1).view(1, -1, 2)


def _patch_warp(H, uv):
    B, P = uv.shape[:2]
    homo = torch.cat((uv, torch.ones((B, P, 1), device=uv.device, dtype=uv.dtype)), dim=-1)
    g = torch.einsum("bik,bpk->bpi", H.view(B, 3, 3), homo).reshape(B, P, 3)
    return g[..., :2] / (g[..., 2:] + 1e-10)


def _patch_gradient(patch, patch_size):
    patch = patch.view(-1, 1, patch_size, patch_size)
    sx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=patch.dtype, device=patch.device).view(1, 1, 3, 3)
    gx, gy = F.conv2d(patch, sx, padding=1), F.conv2d(patch, sx.transpose(-1, -2), padding=1)
    return torch.sqrt(gx ** 2 + gy ** 2 + 1e-6)


def _loss_ncc(ref, nea, std_mask=False):
    bs, tps = nea.shape
    ps = int(np.sqrt(tps))
    filt = torch.ones(1, 1, ps, ps, device=ref.device, dtype=ref.dtype)
    pad = ps // 2
    v = lambda t: F.conv2d(t.view(bs, 1, ps, ps), filt, stride=1, padding=pad)[:, :, pad, pad]
    ref_sum, nea_sum, ref2_sum, nea2_sum, rn_sum = v(ref), v(nea), v(ref.pow(2)), v(nea.pow(2)), v(ref * nea)
    ref_avg, nea_avg = ref_sum / tps, nea_sum / tps
    cross = rn_sum - nea_avg * ref_sum
    ref_var = ref2_sum - ref_avg * ref_sum
    nea_var = nea2_sum - nea_avg * nea_sum
    ncc = torch.clamp(1 - cross * cross / (ref_var * nea_var + 1e-8), 0.0, 2.0)
    ncc = torch.mean(ncc, dim=1, keepdim=True)
    return ncc, (torch.sqrt(ref_var) < 0.01) if std_mask else (ncc < 0.9)


def patch_ncc_torch(pixels, normals, dists, ref_cam, near_cam, ncc_scale, patch, dtype=torch.float32, roughness=False):
    """`dtype=torch.float64` evaluates the same formulation in double precision (the arbiter in the tests: the variances
    are differences of nearly equal sums, so two fp32 evaluations of a low-texture patch legitimately differ by ~1e-2)."""
    dev = pixels.device
    c = lambda t: t.to(dtype)
    pixels, normals, dists = c(pixels), c(normals), c(dists)
    ori = pixels.reshape(-1, 1, 2) / ncc_scale + c(_patch_offsets(patch, dev))
    gray = c(ref_cam.gray_image)
    h, w = gray.squeeze().shape
    pp = ori.clone()
    pp[:, :, 0] = 2 * pp[:, :, 0] / (w - 1) - 1.0
    pp[:, :, 1] = 2 * pp[:, :, 1] / (h - 1) - 1.0
    tps = (patch * 2 + 1) ** 2
    ref_val = F.grid_sample(gray.unsqueeze(1), pp.view(1, -1, 1, 2), align_corners=True).reshape(-1, tps)
    Vr, Vn = c(ref_cam.world_view_transform), c(near_cam.world_view_transform)
    rn_R = Vn[:3, :3].transpose(-1, -2) @ Vr[:3, :3]
    rn_t = -rn_R @ Vr[3, :3] + Vn[3, :3]
    n = dists.shape[0]
    H = rn_R[None] - torch.matmul(rn_t[None, :, None].expand(n, 3, 1), normals[:, :, None].expand(n, 3, 1).permute(0, 2, 1)) / dists[..., None, None]
    H = torch.matmul(c(near_cam.get_K(ncc_scale))[None].expand(n, 3, 3), H)
    H = H @ c(ref_cam.get_inv_K(ncc_scale))
    grid = _patch_warp(H.reshape(-1, 3, 3), ori)
    gx = 2 * grid[:, :, 0] / (w - 1) - 1.0
    gy = 2 * grid[:, :, 1] / (h - 1) - 1.0
    samp = F.grid_sample(c(near_cam.gray_image)[None], torch.stack((gx, gy), dim=-1).reshape(1, -1, 1, 2), align_corners=True).reshape(-1, tps)
    if roughness:  # utils/loss_utils.py:204-211
        ps = patch * 2 + 1
        ncc_grad, _ = _loss_ncc(_patch_gradient(ref_val, ps).view(-1, tps), _patch_gradient(samp, ps).view(-1, tps))
        ncc_gray, std_mask = _loss_ncc(ref_val, samp, std_mask=True)
        return ncc_gray, ncc_grad, std_mask
    return _loss_ncc(ref_val, samp)


# ---------------------------------------------------------------- scene side (scene/__init__.py:125-141, 150-204)
class MultiViewScene:
    def __init__(self, cameras, gt_images, gaussians, opt=MultiViewParams, ncc_scale=1.0):
        self.cameras, self.gaussians, self.ncc_scale = cameras, gaussians, float(ncc_scale)
        cam = cameras[0]
        ix, iy = torch.meshgrid(torch.arange(cam.image_width), torch.arange(cam.image_height), indexing="xy")
        self.pixels = torch.stack([ix, iy], dim=-1).float().to(cam.device)
        for c, img in zip(cameras, gt_images):
            if self.ncc_scale != 1.0:
                res = (int(c.image_height / self.ncc_scale), int(c.image_width / self.ncc_scale))
                img = F.interpolate(img[None], size=res, mode="bilinear", align_corners=False, antialias=True)[0]
            c.gray_image = (img[0:1] * 0.299 + img[1:2] * 0.587 + img[2:3] * 0.114).contiguous()
        self.populate_neighbor_cameras(opt)

    def getTrainCameras(self):
        return self.cameras

    def populate_neighbor_cameras(self, opt):
        centres = torch.stack([c.camera_center for c in self.cameras], dim=0)
        rays = F.normalize(torch.stack([torch.tensor(c.R, dtype=torch.float32)[:3, 2] for c in self.cameras], dim=0).to(centres.device), dim=-1)
        dist = torch.norm(centres[:, None] - centres[None], dim=-1).cpu().numpy()
        ang = (torch.arccos(torch.sum(rays[:, None] * rays[None], dim=-1)) * 180 / 3.14159).cpu().numpy()
        for i, cam in enumerate(self.cameras):
            order = np.lexsort((ang[i], dist[i]))
            ok = (ang[i][order] <= opt.multi_view_max_angle) & (dist[i][order] > opt.multi_view_min_dist) & (dist[i][order] < opt.multi_view_max_dist)
            cam.nearest_indices = [int(k) for k in order[ok][:opt.multi_view_num]]
            nb = order[(ang[i][order] <= opt.nearby_cam_max_angle) & (ang[i][order] >= opt.nearby_cam_min_angle) &
                       (dist[i][order] >= opt.nearby_cam_min_dist) & (dist[i][order] <= opt.nearby_cam_max_dist)]
            k = min(opt.nearby_cam_num, len(nb))
            cam.nearby_indices = [int(nb[j]) for j in np.round(np.linspace(0, len(nb) - 1, k)).astype(int)] if k > 0 else []


# ---------------------------------------------------------------- the loss (utils/loss_utils.py:245-349, 351-449)
def _mm3(pts, M):
    """pts (N, 3) @ M (3, 3) as three broadcast multiply-adds: on ROCm a K = 3 matmul goes to the BLAS library and costs
    ~20 ms for the 2M pixels of a 1080p view (the same trap as in render()); this is three small kernels."""
    return pts[:, 0:1] * M[0] + pts[:, 1:2] * M[1] + pts[:, 2:3] * M[2]


def _get_points_from_depth(camera, depth_map):
    pts = (camera.get_rays() * depth_map.squeeze()[..., None]).reshape(-1, 3)
    R = torch.tensor(camera.R, dtype=torch.float32, device=pts.device)
    T = torch.tensor(camera.T, dtype=torch.float32, device=pts.device)
    return _mm3(pts - T, R.transpose(-1, -2))


def _sample_depth_normal(cam_points, camera, render_pkg, fused=True):
    W, H = int(camera.image_width), int(camera.image_height)
    proj = torch.stack([cam_points[:, 0] * camera.Fx / cam_points[:, 2] + camera.Cx,
                        cam_points[:, 1] * camera.Fy / cam_points[:, 2] + camera.Cy], dim=-1).float()
    valid = (proj[:, 0] > 0) & (proj[:, 0] < W) & (proj[:, 1] > 0) & (proj[:, 1] < H) & (cam_points[:, 2] > 0.1)
    grid = torch.stack([proj[:, 0] / ((W - 1) / 2) - 1, proj[:, 1] / ((H - 1) / 2) - 1], dim=-1)
    if fused and grid.is_cuda:  # one 4-channel lookup, HIP forward and backward (include/gs2m_mvs.h)
        zn = grid_sample_border(torch.cat((render_pkg["depth_map"], render_pkg["normal_map"]), dim=0), grid)
        map_z, map_n = zn[:, 0], zn[:, 1:4]
    else:
        g4 = grid.view(1, -1, 1, 2)
        map_z = F.grid_sample(render_pkg["depth_map"][None], g4, mode="bilinear", padding_mode="border", align_corners=True)[0, 0, :, 0]
        map_n = F.grid_sample(render_pkg["normal_map"][None], g4, mode="bilinear", padding_mode="border", align_corners=True)[0, :, :, 0].permute(1, 0)
    return map_z, map_n / (map_n.norm(dim=1, keepdim=True) + 1e-8), valid


def _reproject_points(from_camera, to_camera, points, sampled_depth):
    pts = points / points[:, 2:3] * sampled_depth[..., None]
    R = torch.tensor(from_camera.R, dtype=torch.float32, device=pts.device)
    T = torch.tensor(from_camera.T, dtype=torch.float32, device=pts.device)
    pts = _mm3(pts - T, R.transpose(-1, -2))
    pts = _mm3(pts, to_camera.world_view_transform[:3, :3]) + to_camera.world_view_transform[3, :3]
    return torch.stack([pts[:, 0] * to_camera.Fx / pts[:, 2] + to_camera.Cx, pts[:, 1] * to_camera.Fy / pts[:, 2] + to_camera.Cy], dim=-1).float()


def _sample_normal_map(pixels, normal_map, fused=True):
    """The reference samples the normal map bilinearly AT THE PIXEL CENTRES (utils/loss_utils.py:428-449): with align_corners that
    is the pixel itself, weight 1 -- and a 2M-point grid_sample whose backward costs 20 ms here.  fused: the reshape it equals."""
    H, W = pixels.shape[:2]
    if fused:
        return normal_map.reshape(normal_map.shape[0], -1).permute(1, 0)
    p = pixels.view(-1, 2)
    grid = torch.stack([p[:, 0] / ((W - 1) / 2) - 1, p[:, 1] / ((H - 1) / 2) - 1], dim=-1).view(1, -1, 1, 2)
    return F.grid_sample(normal_map.unsqueeze(0), grid, mode="bilinear", padding_mode="border", align_corners=True)[0, :, :, 0].permute(1, 0)


class _MVTake(torch.autograd.Function):
    """What the sampled pixels hand the NCC kernel, in one launch (include/gs2m_loss.h: gs2m_mv_take_*): idx (n) distinct flat pixel
    indices, local_normal_map (3,H,W), distance_map (H,W) or (1,H,W), w_map (H,W) or None -> pixels (n,2), normals (n,3), dists (n), w (n).
    Gradients to the two maps (a scatter into zeros: the indices are distinct)."""

    @staticmethod
    def forward(ctx, idx, normal_map, dist_map, w_map):
        normal_map, dist_map = normal_map.contiguous().float(), dist_map.contiguous().float()
        H, W = normal_map.shape[-2:]
        n, dev = idx.numel(), normal_map.device
        idx = idx.contiguous().long()
        pixels = torch.empty((n, 2), dtype=torch.float32, device=dev)
        normals = torch.empty((n, 3), dtype=torch.float32, device=dev)
        dists = torch.empty((n,), dtype=torch.float32, device=dev)
        w = torch.empty((n,), dtype=torch.float32, device=dev)
        wm = None if w_map is None else w_map.contiguous().float()
        with _native.device_guard(dev):
            _native.check(_native.lib().gs2m_mv_take_forward(n, idx.data_ptr(), W, H, normal_map.data_ptr(), dist_map.data_ptr(),
                                                             None if wm is None else wm.data_ptr(), pixels.data_ptr(), normals.data_ptr(),
                                                             dists.data_ptr(), w.data_ptr(), C.c_void_p(_native.stream_ptr(dev))), "gs2m_mv_take_forward")
        ctx.save_for_backward(idx)
        ctx.shapes = (normal_map.shape, dist_map.shape, W, H)
        ctx.mark_non_differentiable(pixels, w)
        return pixels, normals, dists, w

    @staticmethod
    def backward(ctx, _gp, g_normals, g_dists, _gw):
        (idx,) = ctx.saved_tensors
        nshape, dshape, W, H = ctx.shapes
        dev = g_normals.device
        maps = torch.zeros((4, H, W), dtype=torch.float32, device=dev)  # one fill for both maps
        with _native.device_guard(dev):
            _native.check(_native.lib().gs2m_mv_take_backward(idx.numel(), idx.data_ptr(), W, H, g_normals.contiguous().float().data_ptr(),
                                                              g_dists.contiguous().float().data_ptr(), maps[:3].data_ptr(), maps[3].data_ptr(),
                                                              C.c_void_p(_native.stream_ptr(dev))), "gs2m_mv_take_backward")
        return None, maps[:3].reshape(nshape), maps[3].reshape(dshape), None


class _NCCTail(torch.autograd.Function):
    """sum(ncc w [ncc < 0.9]) / max(#[ncc < 0.9], 1) (utils/loss_utils.py:345-349), one launch each way; w carries no gradient."""

    @staticmethod
    def forward(ctx, ncc, w):
        import gs2m_losses
        ncc, w = ncc.contiguous().float().reshape(-1), w.contiguous().float().reshape(-1)
        dev = ncc.device
        out = torch.empty(2, dtype=torch.float32, device=dev)
        with _native.device_guard(dev):
            _native.check(_native.lib().gs2m_ncc_tail_forward(ncc.numel(), ncc.data_ptr(), w.data_ptr(), out.data_ptr(),
                                                              gs2m_losses._workspace(dev).data_ptr(), C.c_void_p(_native.stream_ptr(dev))), "gs2m_ncc_tail_forward")
        ctx.save_for_backward(ncc, w, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        ncc, w, out = ctx.saved_tensors
        d = torch.empty_like(ncc)
        with _native.device_guard(ncc.device):
            _native.check(_native.lib().gs2m_ncc_tail_backward(ncc.numel(), ncc.data_ptr(), w.data_ptr(), out.data_ptr(), g.contiguous().float().data_ptr(),
                                                               d.data_ptr(), C.c_void_p(_native.stream_ptr(ncc.device))), "gs2m_ncc_tail_backward")
        return d.reshape(-1, 1), None


class _TakeDistinct(torch.autograd.Function):
    """x[idx] along dim 0 for DISTINCT indices (random_subset's): the backward is a plain scatter into zeros.  torch's indexing backward
    cannot know the indices are distinct and sorts them first to add duplicates up deterministically: two 100 k-key merge sorts per
    iteration for the normals and distances the NCC patches start from."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return x[idx]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        out.index_copy_(0, idx, g.contiguous())
        return out, None


def take_distinct(x, idx):
    return _TakeDistinct.apply(x, idx) if x.requires_grad else x[idx]


_subset_scratch = {}


def _subset_device(flat, k, rng):
    """random_subset's two steps as four launches (include/gs2m_loss.h: gs2m_subset_thin / gs2m_subset_remove) with counter-based random
    numbers; the seeds come from `rng` (Python's `random`: seeded by the training run, no device generator state).  -> indices, or None
    when the rare plain path is needed (the thinning came out short, or most survivors would have to be removed)."""
    dev = flat.device
    n = flat.shape[0]
    cap = int(k + 12.0 * math.sqrt(k) + 64)
    key = (dev, _native.stream_ptr(dev))
    sc = _subset_scratch.get(key)
    if sc is None or sc[0].shape[0] < cap or sc[2].shape[0] < (n + 1023) // 1024:
        sc = (torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(2, dtype=torch.int32, device=dev),
              torch.empty((n + 1023) // 1024 + 1, dtype=torch.int32, device=dev), torch.empty(2, dtype=torch.int32).pin_memory())
        _subset_scratch[key] = sc
    idx, counts, blocks, host = sc
    s1, s2 = rng.getrandbits(63), rng.getrandbits(63)
    with _native.device_guard(dev):
        _native.check(_native.lib().gs2m_subset_thin(n, flat.data_ptr(), int(k), s1, idx.data_ptr(), cap, counts.data_ptr(), blocks.data_ptr(),
                                                     C.c_void_p(_native.stream_ptr(dev))), "gs2m_subset_thin")
        host.copy_(counts, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()  # (the one host wait `nonzero` always had)
        total, m = int(host[0]), int(host[1])
        if m > cap or (m < k and total > m):
            return None
        if m <= k:
            return idx[:m].clone()
        if 4 * (m - k) > m:
            return None
        out = torch.empty(int(k), dtype=torch.int64, device=dev)
        _native.check(_native.lib().gs2m_subset_remove(m, int(k), s2, idx.data_ptr(), out.data_ptr(), C.c_void_p(_native.stream_ptr(dev))),
                      "gs2m_subset_remove")
    return out


def random_subset(mask, k, rng=random):
    """-> indices of a random subset of exactly min(k, count) set elements of the flat bool `mask`, every set element equally likely --
    what the reference draws with `idx[torch.randperm(idx.numel())[:k]]` (utils/loss_utils.py:283-286, 172-175), without sorting one
    random key per VALID PIXEL (450 k keys per iteration at DTU's size: 0.18 ms of merge-sort kernels; the sort's ~12 launches cost
    nearly as much for 100 k keys).  Two steps, no sort: the mask is thinned to k + 4 sqrt(k) expected survivors (one compare
    against uniform numbers), then the ~1 % in excess are removed one per stratum of the survivor list (position (j + U_j) m / e for
    the j-th of e removals: distinct, increasing, uniform over the list), and the kept positions follow from a searchsorted.  Not the
    uniform distribution over k-subsets (the removals are stratified), but uniform inclusion probabilities and exactly k samples, in
    element order.  On the GPU: four launches of this repository's kernels (`_subset_device`; round 5: ~27 framework operators);
    GS2M_SUBSET_TORCH=1 or a CPU mask: the PyTorch formulation below.  (One host wait, as `nonzero` always had.)"""
    flat = mask.reshape(-1)
    if flat.is_cuda and flat.dtype == torch.bool and flat.is_contiguous() and flat.shape[0] > 0 and k >= 1 and os.environ.get("GS2M_SUBSET_TORCH") is None:
        out = _subset_device(flat, k, rng)
        if out is not None:
            return out
    n = flat.sum()
    p = ((k + 4.0 * math.sqrt(k)) / n.clamp(min=1).to(torch.float32)).clamp(max=1.0)
    idx = torch.nonzero(flat & (torch.rand(flat.shape[0], device=flat.device) < p)).squeeze(1)
    m = idx.numel()
    if m > k:
        e = m - k
        if 4 * e > m:  # many to remove (a small k out of a short list): the plain way
            return idx[torch.randperm(m, device=idx.device)[:k]]
        j = torch.arange(e, device=idx.device, dtype=torch.float64)
        removed = ((j + torch.rand(e, device=idx.device, dtype=torch.float64)) * (m / e)).long().clamp_(max=m - 1)  # strictly increasing: m / e >= 4
        out = torch.arange(k, device=idx.device)
        # the j-th kept element sits at old position j + #{i : removed_i - i <= j}
        return idx[out + torch.searchsorted(removed - torch.arange(e, device=idx.device), out, right=True)]
    if m < k and int(n) > m:  # the thinning came out short (4 sigma): the plain way
        idx = torch.nonzero(flat).squeeze(1)
        if idx.numel() > k:
            idx = idx[torch.randperm(idx.numel(), device=idx.device)[:k]]
    return idx


def multi_view_loss(scene, viewpoint_cam, opt, render_pkg, pipe, bg_color, material_stage, render_fn, fused=True, rng=random):
    cams = scene.getTrainCameras()
    if len(viewpoint_cam.nearest_indices) == 0:
        return 0.0
    near = cams[rng.sample(viewpoint_cam.nearest_indices, 1)[0]]
    # only the neighbour's depth and normal maps enter the loss: this repository's render() takes `shade=False` (no SH evaluation, no
    # dL/dSH); any other render function (the reference's signature) is called as the reference calls it
    extra = {"shade": False} if (getattr(render_fn, "__module__", "") == "gaussian_renderer" and os.environ.get("GS2M_SHADE_NEIGHBOUR") is None) else {}
    near_pkg = render_fn(near, scene.gaussians, pipe, bg_color, geometry_stage=True, material_stage=False, sobel_normal=False, **extra)
    if fused and os.environ.get("GS2M_MV_GEO_TORCH") is None:  # (env: debugging aid, the op-by-op geometric chain with the fused NCC)
        pixel_noise, angle, valid = mv_geo(render_pkg["depth_map"], render_pkg["normal_map"], near_pkg["depth_map"], near_pkg["normal_map"],
                                           viewpoint_cam, near, opt.mv_occlusion_threshold)
    else:
        pixel_noise, angle, valid = mv_geo_torch(render_pkg["depth_map"], render_pkg["normal_map"], near_pkg["depth_map"], near_pkg["normal_map"],
                                                 viewpoint_cam, near, opt.mv_occlusion_threshold, scene.pixels)
    fused_geo = fused and pixel_noise.is_cuda and os.environ.get("GS2M_MV_GEO_TORCH") is None
    if fused_geo:  # the masked means, the masks and the NCC weights in one pass over the frame (gs2m_mv_geo_loss_*)
        w_geo_loss, pixel_valid, w_ncc_map = mv_geo_loss(pixel_noise, angle, valid, opt)
    else:
        angle_valid = valid & (angle < opt.mv_angle_threshold * torch.pi / 180.0)
        pixel_valid = valid & (pixel_noise < 1.0)
        geo_w = torch.where(pixel_valid, torch.exp(-pixel_noise * opt.mv_geo_weight_decay), 0.0).detach()
        # masked means instead of boolean-mask gathers (same values up to the summation order; no nonzero / host sync)
        pixel_loss = torch.where(pixel_valid, geo_w * pixel_noise, 0.0).sum() / pixel_valid.sum().clamp(min=1)  # where, not *: 0 * inf = NaN
        angle_loss = (geo_w * (opt.mv_angle_factor * angle) * angle_valid).sum() / angle_valid.sum().clamp(min=1)
        w_geo_loss = opt.multi_view_geo_weight * (pixel_loss + angle_loss)
        w_ncc_map = None
    if pipe.z_depth:
        return w_geo_loss
    with torch.no_grad():
        idx = random_subset(pixel_valid, opt.multi_view_sample_num)
        if idx.numel() == 0:
            return w_geo_loss
        w_map = w_ncc_map if w_ncc_map is not None else torch.where(pixel_valid, torch.exp(-pixel_noise), 0.0)
    if fused and w_map.is_cuda:  # the samples' inputs in one launch, the masked mean in one (include/gs2m_loss.h: gs2m_mv_take_*, gs2m_ncc_tail_*)
        pixels, local_n, local_d, w_ncc = _MVTake.apply(idx, render_pkg["local_normal_map"], render_pkg["distance_map"], w_map)
        if material_stage:
            w_ncc = w_ncc * (render_pkg["roughness_map"].detach().squeeze().clamp(0, 1) ** 2.0).reshape(-1)[idx]
        M, b, Kinv = _homography_constants(viewpoint_cam, near, scene.ncc_scale)
        ncc = _PatchNCC.apply(pixels, local_n, local_d, viewpoint_cam.gray_image, near.gray_image, M, b, Kinv, scene.ncc_scale, opt.multi_view_patch_size)
        return w_geo_loss + opt.multi_view_ncc_weight * _NCCTail.apply(ncc, w_ncc)
    with torch.no_grad():
        w_ncc = w_map.reshape(-1)[idx]
        if material_stage:
            w_ncc = w_ncc * (render_pkg["roughness_map"].squeeze().clamp(0, 1) ** 2.0).reshape(-1)[idx]
        pixels = scene.pixels.reshape(-1, 2)[idx]
    local_n = take_distinct(render_pkg["local_normal_map"].permute(1, 2, 0).reshape(-1, 3), idx)
    local_d = take_distinct(render_pkg["distance_map"].reshape(-1), idx)
    ncc, mask = (patch_ncc if fused else patch_ncc_torch)(pixels, local_n, local_d, viewpoint_cam, near, scene.ncc_scale, opt.multi_view_patch_size)
    m = mask.reshape(-1)
    ncc_loss = (ncc.reshape(-1) * w_ncc * m).sum() / m.sum().clamp(min=1)
    return w_geo_loss + opt.multi_view_ncc_weight * ncc_loss


def roughness_loss(scene, viewpoint_cam, opt, render_pkg, pipe, bg_color, render_fn, fused=True, rng=random):
    """utils/loss_utils.py:138-230: where a NEARBY view (10-60 degrees away) still correlates photometrically the surface
    is diffuse and its roughness is pushed up, where it does not (a highlight moved) it is pushed down; only the sampled
    roughness values carry gradient."""
    if pipe.z_depth or len(viewpoint_cam.nearby_indices) == 0:
        return 0.0
    near = scene.getTrainCameras()[rng.sample(viewpoint_cam.nearby_indices, 1)[0]]
    with torch.no_grad():
        pts = _get_points_from_depth(viewpoint_cam, render_pkg["depth_map"])
        pts_near = _mm3(pts, near.world_view_transform[:3, :3]) + near.world_view_transform[3, :3]
        near_pkg = render_fn(near, scene.gaussians, pipe, bg_color, geometry_stage=True, material_stage=False, sobel_normal=False)
        map_z, _, valid = _sample_depth_normal(pts_near, near, near_pkg)
        valid = valid & (pts_near[:, 2] - map_z <= opt.mv_occlusion_threshold)
        idx = random_subset(valid, opt.multi_view_sample_num)
        if idx.numel() == 0:
            return 0.0
        pixels = scene.pixels.reshape(-1, 2)[idx]
        local_n = render_pkg["local_normal_map"].permute(1, 2, 0).reshape(-1, 3)[idx]
        local_d = render_pkg["distance_map"].reshape(-1)[idx]
        if fused:
            ncc_gray, ncc_grad, std_mask = patch_ncc_roughness(pixels, local_n, local_d, viewpoint_cam, near, scene.ncc_scale, opt.multi_view_patch_size)
        else:
            ncc_gray, ncc_grad, std_mask = patch_ncc_torch(pixels, local_n, local_d, viewpoint_cam, near, scene.ncc_scale, opt.multi_view_patch_size, roughness=True)
        err = torch.tanh(8.0 * (torch.where(std_mask, ncc_grad, ncc_gray).reshape(-1) - opt.reflection_threshold))
    rough = take_distinct(render_pkg["roughness_map"].reshape(-1), idx)       # grid_sample at integer pixel positions = the pixel itself
    m = ((err < 0.0) & (rough <= 0.8).detach()) | ((err > 0.0) & (rough > 0.08).detach())
    return (err * rough * m).sum() / m.sum().clamp(min=1)
