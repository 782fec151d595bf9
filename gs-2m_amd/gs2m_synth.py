"""Seeded synthetic cameras / Gaussians / upstream gradients.

Used by bench.py and by the tests so that every run sees the same workload.  Camera
matrices are built the way the reference builds them (scene/cameras.py:64-67 on top of
utils/graphics_utils.py:38-71): `viewmatrix` is the transposed world-to-camera matrix,
`projmatrix` = viewmatrix @ P^T, `campos` = inverse(viewmatrix)[3, :3].
"""
import math

import numpy as np
import torch


def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """W2C 4x4 from the reference's camera convention (R is the transposed W2C rotation).
    Restates utils/graphics_utils.py:38-49 (getWorld2View2)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).transpose()
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.float32(np.linalg.inv(C2W))


def projection_matrix(znear, zfar, fovX, fovY):
    """Restates utils/graphics_utils.py:51-71 (getProjectionMatrix), z_sign = +1."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, fx=None, fy=None, R=None, T=None, znear=0.01, zfar=100.0):
    """fx defaults to 1200 px at W=1920 (tanfovx 0.8), scaled with W; fy = fx."""
    if fx is None:
        fx = 1200.0 * W / 1920.0
    if fy is None:
        fy = fx
    fovx = 2 * math.atan(W / (2 * fx))
    fovy = 2 * math.atan(H / (2 * fy))
    R = np.eye(3) if R is None else np.asarray(R)
    T = np.zeros(3) if T is None else np.asarray(T)
    view = torch.tensor(world2view(R, T)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    campos = view.inverse()[3, :3].contiguous()
    return dict(W=W, H=H, fx=fx, fy=fy, FoVx=fovx, FoVy=fovy, tanfovx=math.tan(fovx * 0.5),
                tanfovy=math.tan(fovy * 0.5), viewmatrix=view, projmatrix=full, campos=campos,
                znear=znear, zfar=zfar, R=np.asarray(R, dtype=np.float64), T=np.asarray(T, dtype=np.float64))


def look_at_camera(W, H, eye, target, up=(0.0, -1.0, 0.0), **kw):
    """Camera at `eye` looking at `target` (+z forward, +y down as in COLMAP)."""
    eye = np.asarray(eye, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    down = -np.asarray(up, dtype=np.float64)
    right = np.cross(down, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rc2w = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world
    Rw2c = Rc2w.T
    T = -Rw2c @ eye
    return make_camera(W, H, R=Rw2c.T, T=T, **kw)  # reference stores R transposed


def orbit_cameras(n, W, H, radius=6.0, centre=(0.0, 0.0, 6.0), **kw):
    """n cameras on a circle around the cloud centre (config C5)."""
    cams = []
    for i in range(n):
        a = 2 * math.pi * i / n
        eye = (centre[0] + radius * math.sin(a), centre[1], centre[2] - radius * math.cos(a))
        cams.append(look_at_camera(W, H, eye, centre, **kw))
    return cams


def make_gaussians(P, cam, seed=0, sh_degree=3, behind_frac=0.01, scale_lo=0.002, scale_hi=0.02,
                   device="cpu"):
    """SURVEY.md 8(d) distribution, relative to the identity-pose camera frustum."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    z = 2.0 + 8.0 * u(P)
    nb = int(P * behind_frac)
    if nb > 0:
        z[:nb] = -1.0 + 1.2 * u(nb)  # U(-1, 0.2): exercised by the near-plane cull
    x = z.abs().clamp_min(0.5) * cam["tanfovx"] * (2.2 * u(P) - 1.1)
    y = z.abs().clamp_min(0.5) * cam["tanfovy"] * (2.2 * u(P) - 1.1)
    means3D = torch.stack([x, y, z], dim=1)
    scales = torch.exp(math.log(scale_lo) + (math.log(scale_hi) - math.log(scale_lo)) * u(P, 3))
    rotations = torch.nn.functional.normalize(n(P, 4), dim=1)
    opacities = 0.05 + 0.9 * u(P, 1)
    M = (sh_degree + 1) ** 2 if sh_degree >= 0 else 0
    shs = torch.cat([n(P, 1, 3), 0.1 * n(P, 15, 3)], dim=1)[:, :max(M, 1)].contiguous()
    if M < 16:
        shs = torch.cat([shs, torch.zeros(P, 16 - shs.shape[1], 3)], dim=1)
    normals = torch.nn.functional.normalize(n(P, 3), dim=1)
    features = torch.cat([torch.ones(P, 1), 1.0 + 9.0 * u(P, 1), normals, u(P, 3), u(P, 1), u(P, 1)], dim=1)
    out = dict(means3D=means3D, scales=scales, rotations=rotations, opacities=opacities, shs=shs,
               features=features)
    return {k: v.float().contiguous().to(device) for k, v in out.items()}


def make_upstream_grads(H, W, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed + 12345)
    Gc = torch.randn(3, H, W, generator=g)
    Gb = torch.randn(10, H, W, generator=g)
    return Gc.to(device), Gb.to(device)


def algo_bytes(P, V, R, N, Tn, fc, M=16):
    """Algorithmic HBM bytes per view; returns dict of stage byte counts."""
    sh = 12 * M
    pre_f = 44 * P + sh * V + 8 * P + 67 * V
    scan = 8 * P
    keygen = 8 * P + 12 * V + 12 * R
    sort = 24 * R
    ranges = 8 * R + 8 * Tn
    blend_f = 8 * Tn + (40 + 4 * fc) * R + (20 + 4 * fc) * N
    blend_b = 8 * Tn + (84 + 8 * fc) * R + (20 + 4 * fc) * N
    gauss_b = 4 * P + (115 + sh) * V + (64 + sh) * V
    fwd = pre_f + scan + keygen + sort + ranges + blend_f
    bwd = blend_b + gauss_b
    return dict(preprocess=pre_f, scan=scan, keygen=keygen, sort=sort, ranges=ranges, blend_fwd=blend_f,
                blend_bwd=blend_b, gaussian_bwd=gauss_b, forward=fwd, backward=bwd, total=fwd + bwd)


def make_surface_scene(n, seed=0, centre=(0.0, 0.0, 6.0), sh_degree=3, detail=0.0, splat=0.05, grain=0.0):
    """A compact synthetic object for the train.py-level configuration (SURVEY.md 8(d) C4 substitute: no dataset on the
    GPU box): n flat Gaussians on a unit-and-a-half sphere plus a ground disc, smoothly varying colours, normals
    known.  `detail` > 0 adds a fine colour texture of that spatial frequency (radians per unit) on top of the smooth
    field -- a model initialised from a sparse subsample then has to densify to reproduce it, as on a real scan --
    `grain` > 0 adds an independent random colour offset of that standard deviation to every true Gaussian (content at
    the scale of the true splats themselves, the way sensor noise and fine surface texture fill a photograph),
    and `splat` is the in-plane size of the true Gaussians (0.05: the small-scene default; use ~1.5 / sqrt(n) x 20 for
    a dense cover by many small splats).  -> dict(points, normals, colors [0,1], scales, rotations (w,x,y,z),
    opacities, shs (n,16,3))."""
    g = torch.Generator().manual_seed(seed)
    ns = int(n * 0.7)
    d = torch.nn.functional.normalize(torch.randn(ns, 3, generator=g), dim=1)
    sphere = 1.5 * d
    r = 3.0 * torch.sqrt(torch.rand(n - ns, generator=g))
    a = 2 * math.pi * torch.rand(n - ns, generator=g)
    disc = torch.stack([r * torch.cos(a), torch.full_like(r, 1.5), r * torch.sin(a)], dim=1)  # +y is down
    pts = torch.cat([sphere, disc], dim=0)
    normals = torch.cat([d, torch.tensor([0.0, -1.0, 0.0]).expand(n - ns, 3)], dim=0)
    colors = 0.5 + 0.45 * torch.stack([torch.sin(2.1 * pts[:, 0] + 0.3), torch.sin(1.7 * pts[:, 1] + 1.1) * torch.cos(1.3 * pts[:, 2]),
                                       torch.cos(2.3 * pts[:, 2] - 0.4)], dim=1)
    if detail > 0.0:
        tex = torch.sin(detail * pts[:, 0]) * torch.sin(detail * pts[:, 1] + 0.7) * torch.sin(detail * pts[:, 2] + 1.9)
        tex2 = torch.sin(detail * pts[:, 0] + 2.1) * torch.sin(1.3 * detail * pts[:, 2] + 0.3) * torch.cos(0.8 * detail * pts[:, 1])
        colors = 0.5 + (colors - 0.5) * 0.55 + 0.2 * torch.stack([tex, tex2, -tex], dim=1)
    if grain > 0.0:
        colors = colors + grain * torch.randn(n, 3, generator=g)
    # rotation taking +z to the normal: q = normalize(1 + n_z, z x n)
    w = 1.0 + normals[:, 2]
    q = torch.stack([w, -normals[:, 1], normals[:, 0], torch.zeros(n)], dim=1)
    q[w < 1e-6] = torch.tensor([0.0, 1.0, 0.0, 0.0])
    q = torch.nn.functional.normalize(q, dim=1)
    scales = torch.tensor([splat, splat, 0.1 * splat]).expand(n, 3) * (0.7 + 0.6 * torch.rand(n, 1, generator=g))
    shs = torch.zeros(n, (sh_degree + 1) ** 2, 3)
    shs[:, 0] = (colors - 0.5) / 0.28209479177387814
    c = torch.tensor(centre)
    return dict(points=(pts + c).float(), normals=normals.float(), colors=colors.float().clamp(0, 1), scales=scales.float().contiguous(),
                rotations=q.float(), opacities=torch.full((n, 1), 0.95), shs=shs.float())
