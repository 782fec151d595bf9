"""Minimal host-side counterparts of the objects `render()` consumes, on PyTorch-ROCm.

Not a re-implementation of the reference's scene/ package: only the accessors the hot-path
caller touches.  Each function cites what it restates.
"""
import math

import torch
import torch.nn.functional as F

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """utils/sh_utils.py:57-112: sh (..., C, (deg+1)^2), dirs (..., 3) -> (..., C)."""
    assert 0 <= deg <= 3
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5]
                      + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + C2[3] * xz * sh[..., 7]
                      + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                          + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                          + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                          + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def build_rotation(r):
    """utils/general_utils.py:75-98 (device taken from the input instead of a hard-coded 'cuda')."""
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device, dtype=r.dtype)
    r_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r_ * z)
    R[:, 0, 2] = 2 * (x * z + r_ * y)
    R[:, 1, 0] = 2 * (x * y + r_ * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r_ * x)
    R[:, 2, 0] = 2 * (x * z - r_ * y)
    R[:, 2, 1] = 2 * (y * z + r_ * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianParams:
    """Parameter container with the reference GaussianModel's activations and accessors
    (scene/gaussian_model.py:28-55, 113-172)."""

    def __init__(self, xyz, features_dc, features_rest, scaling, rotation, opacity, albedo, roughness, metallic,
                 active_sh_degree=3, max_sh_degree=3):
        self._xyz, self._features_dc, self._features_rest = xyz, features_dc, features_rest
        self._scaling, self._rotation, self._opacity = scaling, rotation, opacity
        self._albedo, self._roughness, self._metallic = albedo, roughness, metallic
        self.active_sh_degree, self.max_sh_degree = active_sh_degree, max_sh_degree

    @classmethod
    def from_activated(cls, means3D, shs, scales, rotations, opacities, albedo, roughness, metallic, **kw):
        """Build raw parameters whose activations reproduce the given activated values."""
        return cls(means3D, shs[:, :1].contiguous(), shs[:, 1:].contiguous(), torch.log(scales), rotations,
                   inverse_sigmoid(opacities), inverse_sigmoid(albedo), inverse_sigmoid(roughness),
                   inverse_sigmoid(metallic), **kw)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
                self._albedo, self._roughness, self._metallic]

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return F.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_albedo(self):
        return torch.sigmoid(self._albedo)

    @property
    def get_roughness(self):
        return torch.sigmoid(self._roughness)

    @property
    def get_metallic(self):
        return torch.sigmoid(self._metallic)

    def get_covariance(self, scaling_modifier=1):
        """build_covariance_from_scaling_rotation (scene/gaussian_model.py:29-34, utils/general_utils.py:61-74, 100-111)."""
        s = scaling_modifier * self.get_scaling
        L = build_rotation(self._rotation) @ torch.diag_embed(s)
        cov = L @ L.transpose(1, 2)
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)

    def get_normals(self, camera_center):
        """scene/gaussian_model.py:146-160: min-scale axis of R, flipped to face the camera."""
        scales = self.get_scaling
        min_axis_idx = torch.argmin(scales, dim=-1, keepdim=True)
        min_axes = torch.zeros_like(scales).scatter(1, min_axis_idx, 1)
        rotations = build_rotation(self.get_rotation)
        normals = torch.bmm(rotations, min_axes.unsqueeze(-1)).squeeze(-1)
        view_dirs = camera_center[None] - self.get_xyz
        flip_mask = torch.sum(normals * view_dirs, dim=-1) < 0.0
        normals = torch.where(flip_mask[:, None], -normals, normals)
        return normals / normals.norm(dim=1, keepdim=True)


class PipelineParams:
    """arguments/__init__.py:70-79 defaults."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    z_depth = False
    split_sh = True          # this repository's addition: hand _features_dc / _features_rest to the rasterizer unconcatenated
    fused_render_ops = True  # this repository's addition: fused HIP pre/post-processing in render() (gs2m_render_ops)
    fused_activations = True  # ... and the model's activation getters as one launch (needs fused_render_ops)
    fused_loss_tail = True   # ... and, in gs2m_train, the loss terms + densification statistics around D-SSIM as HIP kernels (gs2m_losses)


class Camera:
    """The fields and helpers of scene/cameras.py:19-117 that render() uses, built from a
    gs2m_synth camera dict (matrices exactly as scene/cameras.py:64-67)."""

    def __init__(self, cam, device):
        self.image_width, self.image_height = int(cam["W"]), int(cam["H"])
        self.FoVx, self.FoVy = cam["FoVx"], cam["FoVy"]
        self.Fx, self.Fy = cam["fx"], cam["fy"]
        self.Cx, self.Cy = 0.5 * cam["W"], 0.5 * cam["H"]
        self.znear, self.zfar = cam["znear"], cam["zfar"]
        self.world_view_transform = cam["viewmatrix"].to(device)
        self.full_proj_transform = cam["projmatrix"].to(device)
        self.camera_center = cam["campos"].to(device)
        self.device = device
        self.R, self.T = cam.get("R"), cam.get("T")   # numpy, the reference's convention (R = transposed W2C rotation)
        self.gray_image = None                         # (1, h, w) at the NCC scale, populated by the multi-view scene
        self.nearest_indices, self.nearby_indices = [], []

    def get_K(self, scale=1.0):  # scene/cameras.py:92-97
        return torch.tensor([[self.Fx / scale, 0.0, self.Cx / scale], [0.0, self.Fy / scale, self.Cy / scale], [0.0, 0.0, 1.0]], device=self.device)

    def get_inv_K(self, scale=1.0):  # scene/cameras.py:99-104
        return torch.tensor([[scale / self.Fx, 0.0, -self.Cx / self.Fx], [0.0, scale / self.Fy, -self.Cy / self.Fy], [0.0, 0.0, 1.0]], device=self.device)

    def get_rays(self, scale=1.0):  # scene/cameras.py:72-81 (constant per camera: built once per scale, not per call)
        cache = self.__dict__.setdefault("_rays", {})
        if scale not in cache:
            cache[scale] = self._build_rays(scale)
        return cache[scale]

    def _build_rays(self, scale):
        h, w = int(self.image_height / scale), int(self.image_width / scale)
        u, v = torch.meshgrid(torch.arange(w, device=self.device, dtype=torch.float32),
                              torch.arange(h, device=self.device, dtype=torch.float32), indexing='xy')
        rx = (scale * u - self.Cx / scale) / self.Fx
        ry = (scale * v - self.Cy / scale) / self.Fy
        return torch.stack((rx, ry, torch.ones_like(rx)), dim=-1)

    def get_calib_matrix_nerf(self, scale=1.0):  # scene/cameras.py:83-90
        intrinsic = torch.tensor([[self.Fx / scale, 0, self.Cx / scale], [0, self.Fy / scale, self.Cy / scale],
                                  [0, 0, 1]]).float()
        extrinsic = self.world_view_transform.transpose(0, 1).contiguous()
        return intrinsic, extrinsic


# ---- utils/normal_utils.py:3-72 (Sobel-style normal from a depth image) ----
def _ndc_2_cam(ndc_xyz, intrinsic, W, H):
    inv_scale = torch.tensor([[W - 1, H - 1]], device=ndc_xyz.device)
    cam_z = ndc_xyz[..., 2:3]
    cam_xy = ndc_xyz[..., :2] * inv_scale * cam_z
    cam_xyz = torch.cat([cam_xy, cam_z], dim=-1)
    return cam_xyz @ torch.inverse(intrinsic[0, ...].t())


def _depth2point(depth_image, intrinsic_matrix, extrinsic_matrix, view_space=False):
    H, W = depth_image.shape
    d = depth_image[None, None, None, ...]
    vx = torch.arange(W, dtype=torch.float32, device=d.device) / (W - 1)
    vy = torch.arange(H, dtype=torch.float32, device=d.device) / (H - 1)
    vx, vy = torch.meshgrid(vx, vy, indexing='xy')
    vx = vx[None, None, None, ...].expand(1, 1, 1, -1, -1)
    vy = vy[None, None, None, ...].expand(1, 1, 1, -1, -1)
    ndc_xyz = torch.stack([vx, vy, d], dim=-1).view(1, 1, 1, H, W, 3)
    xyz_cam = _ndc_2_cam(ndc_xyz, intrinsic_matrix[None, ...], W, H).reshape(-1, 3)
    if view_space:
        return xyz_cam
    xyz_world = torch.cat([xyz_cam, torch.ones_like(xyz_cam[..., 0:1])], axis=-1) @ torch.inverse(extrinsic_matrix).transpose(0, 1)
    return xyz_world[..., :3]


def _depth_pcd2normal(xyz):
    hd, wd, _ = xyz.shape
    bottom_point = xyz[..., 2:hd, 1:wd - 1, :]
    top_point = xyz[..., 0:hd - 2, 1:wd - 1, :]
    right_point = xyz[..., 1:hd - 1, 2:wd, :]
    left_point = xyz[..., 1:hd - 1, 0:wd - 2, :]
    xyz_normal = torch.cross(right_point - left_point, top_point - bottom_point, dim=-1)
    xyz_normal = F.normalize(xyz_normal, p=2, dim=-1)
    return F.pad(xyz_normal.permute(2, 0, 1), (1, 1, 1, 1), mode='constant').permute(1, 2, 0)


def normal_from_depth_image(depth, intrinsic_matrix, extrinsic_matrix, view_space=False):
    """depth (H,W) -> normals (H,W,3); utils/normal_utils.py:65-72 with offset=None."""
    xyz = _depth2point(depth, intrinsic_matrix, extrinsic_matrix, view_space).reshape(*depth.shape, 3)
    return _depth_pcd2normal(xyz)
