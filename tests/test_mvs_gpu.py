"""The photometric multi-view term: the fused patch-warp + NCC kernel (include/gs2m_mvs.h) against
the reference's op-by-op formulation (gs2m_mvs.patch_ncc_torch, restating utils/loss_utils.py:303-349, 451-509) and against
geometry: two views of a textured plane must correlate perfectly through the TRUE plane's homography."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plane_scene(W=320, H=200, seed=0, dev="cuda"):
    """Two cameras looking at the textured plane through (0, 0, 6) with normal towards them; grey images are the texture
    evaluated at each pixel's ray / plane intersection (exact, no rasterizer involved)."""
    import gs2m_synth as S
    from gs2m_scene import Camera
    cams = [Camera(S.look_at_camera(W, H, eye, (0.0, 0.0, 6.0), fx=1.1 * W), dev) for eye in ((0.0, 0.0, 0.0), (0.9, -0.3, 0.4))]
    n_w = torch.tensor([0.15, -0.1, -1.0], dtype=torch.float64)
    n_w = n_w / n_w.norm()
    p0 = torch.tensor([0.0, 0.0, 6.0], dtype=torch.float64)
    e1 = torch.linalg.cross(n_w, torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64))
    e1 = e1 / e1.norm()
    e2 = torch.linalg.cross(n_w, e1)

    def tex(u, v):
        return 0.5 + 0.2 * torch.sin(3.1 * u + 0.3) * torch.cos(2.3 * v) + 0.15 * torch.sin(5.7 * v + 1.0) + 0.1 * torch.cos(4.1 * (u + v))

    for c in cams:
        V = c.world_view_transform.double().cpu()
        Rcw, centre = V[:3, :3], c.camera_center.double().cpu()      # x_cam = x_w @ V[:3,:3] + V[3,:3]
        rays_c = c.get_rays().double().cpu().reshape(-1, 3)
        rays_w = rays_c @ Rcw.T
        t = ((p0 - centre) @ n_w) / (rays_w @ n_w)
        X = centre + t[:, None] * rays_w
        c.gray_image = tex((X - p0) @ e1, (X - p0) @ e2).float().reshape(1, H, W).to(dev)
        c.plane_n = (n_w @ Rcw).float().to(dev)                        # camera-space normal (faces the camera: n . X < 0)
        c.plane_d = float(abs((centre - p0) @ n_w))
    return cams


@pytest.mark.parametrize("patch", [3, 1])
def test_true_plane_correlates_and_fused_equals_op_by_op(patch):
    assert torch.cuda.is_available()
    import gs2m_mvs as MV
    ref, near = _plane_scene()
    W, H = ref.image_width, ref.image_height
    g = torch.Generator().manual_seed(1)
    N = 4000
    pixels = torch.stack([torch.rand(N, generator=g) * (W - 40) + 20, torch.rand(N, generator=g) * (H - 40) + 20], dim=-1).cuda()
    pixels[:200] = pixels[:200].round()
    n = ref.plane_n[None].expand(N, 3).contiguous()
    d = torch.full((N,), ref.plane_d, device="cuda")
    assert (n @ torch.tensor([0.0, 0.0, 1.0], device="cuda") < 0).all()
    ncc, mask = MV.patch_ncc(pixels, n, d, ref, near, 1.0, patch)
    # the same surface seen twice: (almost) perfect correlation wherever the warped patch lands inside the other image
    # (3x3 patches hold too little contrast on this smooth texture for the statement to be sharp: checked for 7x7)
    assert ncc.shape == (N, 1)
    worse, _ = MV.patch_ncc(pixels, n, d * 1.08, ref, near, 1.0, patch)           # a wrong plane decorrelates
    if patch == 3:
        assert torch.quantile(ncc, 0.8).item() < 2e-3 and mask.float().mean().item() > 0.8
        assert torch.quantile(worse, 0.5).item() > 10 * torch.quantile(ncc, 0.5).item() + 1e-3
    # fused kernel against the op-by-op formulation, on perturbed planes so that values and gradients are non-trivial
    nn_ = torch.nn.functional.normalize(n + 0.05 * torch.randn(N, 3, generator=g).cuda(), dim=-1).requires_grad_(True)
    dd = (d * (1.0 + 0.03 * torch.randn(N, generator=g).cuda())).requires_grad_(True)
    Gw = torch.rand(N, 1, generator=g).cuda()
    a, ma = MV.patch_ncc(pixels, nn_, dd, ref, near, 1.0, patch)
    (a * Gw).sum().backward()
    ga_n, ga_d = nn_.grad.clone(), dd.grad.clone()
    nn_.grad = dd.grad = None
    b, mb = MV.patch_ncc_torch(pixels, nn_, dd, ref, near, 1.0, patch)
    (b * Gw).sum().backward()
    gb_n, gb_d = nn_.grad.clone(), dd.grad.clone()
    nn_.grad = dd.grad = None
    t, _ = MV.patch_ncc_torch(pixels, nn_, dd, ref, near, 1.0, patch, dtype=torch.float64)   # the arbiter
    (t * Gw.double()).sum().backward()
    ea, eb = (a.double() - t).abs().reshape(-1), (b.double() - t).abs().reshape(-1)
    # both fp32 evaluations sit at the same distance from the double-precision value (low-texture patches: ~1e-2)
    assert ea.mean().item() < 1e-4 and torch.quantile(ea, 0.99).item() < 2e-3
    assert ea.mean().item() < 3 * eb.mean().item() + 1e-6 and ea.max().item() < 3 * eb.max().item() + 1e-3
    assert (ma != mb).float().mean().item() < 5e-3                                 # only samples sitting on the 0.9 threshold
    gt_n, gt_d = nn_.grad.double(), dd.grad.double()
    for x, y, z in ((ga_n, gb_n, gt_n), (ga_d, gb_d, gt_d)):
        assert torch.isfinite(x).all()
        ex, ey = (x.double() - z).norm().item(), (y.double() - z).norm().item()
        assert ex < 3 * ey + 5e-3 * z.norm().item(), (ex, ey, z.norm().item())   # 3x3 patches: D^2 in the denominator


def test_patch_ncc_edge_cases():
    """Patches leaving either image (zero padding), degenerate planes, a down-scaled NCC image."""
    assert torch.cuda.is_available()
    import gs2m_mvs as MV
    ref, near = _plane_scene(W=160, H=100)
    g = torch.Generator().manual_seed(2)
    N = 1500
    pixels = torch.stack([torch.rand(N, generator=g) * 159, torch.rand(N, generator=g) * 99], dim=-1).cuda()   # up to the borders
    n = torch.nn.functional.normalize(ref.plane_n[None] + 0.3 * torch.randn(N, 3, generator=g).cuda(), dim=-1)
    d = ref.plane_d * (0.5 + torch.rand(N, generator=g).cuda())
    d[:5] = 1e-6                                                    # nearly singular homographies
    a, _ = MV.patch_ncc(pixels, n, d, ref, near, 1.0, 3)
    b, _ = MV.patch_ncc_torch(pixels, n, d, ref, near, 1.0, 3)
    ok = torch.isfinite(b.reshape(-1))
    assert torch.isfinite(a.reshape(-1)[ok]).all() and (a.reshape(-1)[ok] - b.reshape(-1)[ok]).abs().max().item() < 2e-3
    assert (a >= 0).all() and (a <= 2).all()
    # NCC scale 2: grey images at half resolution, pixel coordinates still full resolution
    for c in (ref, near):
        c.gray_image = torch.nn.functional.avg_pool2d(c.gray_image[None], 2)[0].contiguous()
    px = pixels[(pixels[:, 0] > 20) & (pixels[:, 0] < 140) & (pixels[:, 1] > 20) & (pixels[:, 1] < 80)]
    nn_, dd = ref.plane_n[None].expand(len(px), 3).contiguous(), torch.full((len(px),), ref.plane_d, device="cuda")
    a2, _ = MV.patch_ncc(px, nn_, dd, ref, near, 2.0, 3)
    b2, _ = MV.patch_ncc_torch(px, nn_, dd, ref, near, 2.0, 3)
    assert (a2 - b2).abs().max().item() < 5e-4


def test_roughness_variant_matches_op_by_op():
    """gs2m_patch_ncc_roughness (grey NCC, Sobel-gradient NCC, low-texture switch) against the op-by-op formulation with its
    conv2d Sobel / sums (utils/loss_utils.py:200-211, 232-238), fp64 as arbiter."""
    assert torch.cuda.is_available()
    import gs2m_mvs as MV
    ref, near = _plane_scene()
    W, H = ref.image_width, ref.image_height
    g = torch.Generator().manual_seed(4)
    N = 3000
    pixels = torch.stack([torch.rand(N, generator=g) * (W - 1), torch.rand(N, generator=g) * (H - 1)], dim=-1).cuda()
    n = torch.nn.functional.normalize(ref.plane_n[None] + 0.05 * torch.randn(N, 3, generator=g).cuda(), dim=-1)
    d = ref.plane_d * (1.0 + 0.03 * torch.randn(N, generator=g).cuda())
    # a flat region in the reference image so that the low-texture switch fires for part of the samples
    ref.gray_image[:, :60, :100] = 0.5 + 0.002 * torch.rand(60, 100, generator=g).cuda()
    a = MV.patch_ncc_roughness(pixels, n, d, ref, near, 1.0, 3)
    b = MV.patch_ncc_torch(pixels, n, d, ref, near, 1.0, 3, roughness=True)
    t = MV.patch_ncc_torch(pixels, n, d, ref, near, 1.0, 3, roughness=True, dtype=torch.float64)
    assert 0.02 < a[2].float().mean().item() < 0.5 and (a[2] != t[2]).float().mean().item() < 5e-3
    for k in (0, 1):
        ea, eb = (a[k].double() - t[k]).abs().reshape(-1), (b[k].double() - t[k]).abs().reshape(-1)
        assert ea.mean().item() < 3 * eb.mean().item() + 2e-5, (k, ea.mean().item(), eb.mean().item())
        assert torch.quantile(ea, 0.99).item() < 5e-3, (k, torch.quantile(ea, 0.99).item())


@pytest.mark.parametrize("C", [1, 4])
def test_grid_sample_border_matches_torch(C):
    assert torch.cuda.is_available()
    import torch.nn.functional as F
    import gs2m_mvs as MV
    g = torch.Generator().manual_seed(C)
    H, W, N = 37, 53, 5000
    img = torch.randn(C, H, W, generator=g).cuda().requires_grad_(True)
    grid = (torch.rand(N, 2, generator=g) * 2.4 - 1.2).cuda()                       # 10 % outside on every side: border clamp
    grid[:6] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [0.0, 0.0], [1.0, -1.0], [-1.3, 0.2], [0.2, 1.3]], device="cuda")
    grid.requires_grad_(True)
    G = torch.randn(N, C, generator=g).cuda()
    ref = F.grid_sample(img[None], grid.view(1, -1, 1, 2), mode="bilinear", padding_mode="border", align_corners=True)[0, :, :, 0].permute(1, 0)
    (ref * G).sum().backward()
    gi, gg = img.grad.clone(), grid.grad.clone()
    img.grad = grid.grad = None
    got = MV.grid_sample_border(img, grid)
    (got * G).sum().backward()
    assert (got - ref).abs().max().item() < 1e-5
    assert (img.grad - gi).abs().max().item() < 1e-4 * max(1.0, gi.abs().max().item())
    assert (grid.grad - gg).abs().max().item() < 1e-4 * max(1.0, gg.abs().max().item())


def test_mv_geo_matches_op_by_op():
    """gs2m_mv_geo_* (pixel_noise, angle, valid per pixel; hand-derived backward to the four maps) against the op-by-op chain of
    utils/loss_utils.py:256-276 with autograd, on maps that keep the chain well conditioned (normals 10-40 degrees apart: away from
    acos' clamp; a smooth depth)."""
    assert torch.cuda.is_available()
    import gs2m_synth as S
    import gs2m_mvs as MV
    from gs2m_scene import Camera
    W, H = 160, 96
    ref = Camera(S.look_at_camera(W, H, (0.0, 0.0, 0.0), (0.0, 0.0, 6.0), fx=1.1 * W), "cuda")
    near = Camera(S.look_at_camera(W, H, (0.5, -0.2, 0.3), (0.0, 0.0, 6.0), fx=1.1 * W), "cuda")
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    mk = lambda t: t.cuda().requires_grad_(True)
    depth = mk((5.5 + 0.4 * torch.sin(2 * xx) * torch.cos(1.5 * yy))[None])
    depth_n = mk((5.3 + 0.4 * torch.cos(1.7 * xx + 0.3) * torch.cos(1.2 * yy))[None])
    base = torch.stack([0.3 * xx, 0.3 * yy, -torch.ones_like(xx)], 0)
    normal = mk(base * (0.6 + torch.rand(1, H, W, generator=g)) + 0.15 * torch.randn(3, H, W, generator=g))        # not unit length
    normal_n = mk(torch.stack([0.3 * xx + 0.4, 0.3 * yy - 0.3, -torch.ones_like(xx)], 0) * 0.8 + 0.05 * torch.randn(3, H, W, generator=g))
    ix, iy = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
    pixels = torch.stack([ix, iy], dim=-1).float().cuda()
    G1, G2 = torch.rand(H * W, generator=g).cuda(), torch.rand(H * W, generator=g).cuda()
    leaves = (depth, normal, depth_n, normal_n)

    def run(fn):
        for t in leaves:
            t.grad = None
        noise, angle, valid = fn()
        ((noise * G1 + angle * G2) * valid).sum().backward()
        return noise.detach(), angle.detach(), valid, [t.grad.clone() for t in leaves]

    a = run(lambda: MV.mv_geo(depth, normal, depth_n, normal_n, ref, near, 5.0))
    b = run(lambda: MV.mv_geo_torch(depth, normal, depth_n, normal_n, ref, near, 5.0, pixels))
    assert 0.3 < b[2].float().mean().item() and (a[2] != b[2]).float().mean().item() < 2e-3
    both = a[2] & b[2]
    assert (a[0] - b[0])[both].abs().max().item() < 2e-3 * max(1.0, b[0][both].abs().max().item())
    assert (a[1] - b[1])[both].abs().max().item() < 1e-4
    assert 0.005 < b[1][both].min().item()                                    # away from acos clamp (angle 0.0014)
    for name, x, y in zip(("depth", "normal", "neighbour depth", "neighbour normal"), a[3], b[3]):
        assert torch.isfinite(x).all(), name
        assert (x - y).norm().item() < 5e-3 * y.norm().item(), (name, (x - y).norm().item(), y.norm().item())
    # the occlusion test and the image border switch pixels off
    a2 = MV.mv_geo(depth, normal, depth_n, normal_n, ref, near, 5e-4)
    assert a2[2].float().mean().item() < a[2].float().mean().item()


@pytest.mark.parametrize("scale", [1.0, 1e-11, 3e7])
def test_scatter_backwards_are_bitwise_reproducible(scale):
    """The bilinear scatters of grid_sample_border's and mv_geo's backwards (include/gs2m_mvs.h, deterministic mode = the default:
    64-bit fixed-point sums scaled by the call's largest contribution): the same bits on every run, whatever the magnitude of the
    upstream gradients, and the float-atomic mode's values to fp32 accumulation accuracy.  Many samples per texel: 200 k samples on a
    37 x 53 image (~100 contributions per texel, in whatever order the hardware retires them)."""
    assert torch.cuda.is_available()
    import gs2m_mvs as MV
    g = torch.Generator().manual_seed(3)
    C, H, W, N = 4, 37, 53, 200_000
    img = torch.randn(C, H, W, generator=g).cuda().requires_grad_(True)
    grid = (torch.rand(N, 2, generator=g) * 2.2 - 1.1).cuda().requires_grad_(True)
    G = (torch.randn(N, C, generator=g) * scale).cuda()

    def run():
        img.grad = grid.grad = None
        (MV.grid_sample_border(img, grid) * G).sum().backward()
        return img.grad.clone(), grid.grad.clone()

    assert MV.is_deterministic()
    a = [run() for _ in range(4)]
    for b in a[1:]:
        assert torch.equal(a[0][0], b[0]) and torch.equal(a[0][1], b[1]), "deterministic mode: identical bits run to run"
    try:
        MV.set_deterministic(False)
        f = run()
    finally:
        MV.set_deterministic(True)
    assert torch.isfinite(a[0][0]).all()
    assert (a[0][0] - f[0]).abs().max().item() <= 2e-5 * f[0].abs().max().item(), "fixed-point sums vs fp32 atomics"
    assert torch.equal(a[0][1], f[1]), "the position gradient is not scattered: the same in both modes"
    # the fixed-point sums against a float64 scatter of the same contributions (torch's op on double inputs)
    import torch.nn.functional as F
    imgd, gridd = img.detach().double().requires_grad_(True), grid.detach().double()
    ref = F.grid_sample(imgd[None], gridd.view(1, -1, 1, 2), mode="bilinear", padding_mode="border", align_corners=True)[0, :, :, 0].permute(1, 0)
    (ref * G.double()).sum().backward()
    err_det = (a[0][0].double() - imgd.grad).abs().max().item()
    err_flt = (f[0].double() - imgd.grad).abs().max().item()
    assert err_det <= 4e-6 * imgd.grad.abs().max().item(), (err_det, imgd.grad.abs().max().item())
    assert err_det <= err_flt * 1.5 + 1e-30, "one rounding per texel instead of one per addition"


def test_mv_geo_backward_is_bitwise_reproducible():
    assert torch.cuda.is_available()
    import gs2m_synth as S
    import gs2m_mvs as MV
    from gs2m_scene import Camera
    W, H = 320, 192
    ref = Camera(S.look_at_camera(W, H, (0.0, 0.0, 0.0), (0.0, 0.0, 6.0), fx=1.1 * W), "cuda")
    near = Camera(S.look_at_camera(W, H, (0.5, -0.2, 0.3), (0.0, 0.0, 6.0), fx=1.1 * W), "cuda")
    g = torch.Generator().manual_seed(1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    mk = lambda t: t.cuda().requires_grad_(True)
    depth = mk((5.5 + 0.4 * torch.sin(2 * xx) * torch.cos(1.5 * yy))[None])
    depth_n = mk((5.3 + 0.4 * torch.cos(1.7 * xx + 0.3) * torch.cos(1.2 * yy))[None])
    normal = mk(torch.stack([0.3 * xx, 0.3 * yy, -torch.ones_like(xx)], 0) + 0.15 * torch.randn(3, H, W, generator=g))
    normal_n = mk(torch.stack([0.3 * xx + 0.4, 0.3 * yy - 0.3, -torch.ones_like(xx)], 0) * 0.8 + 0.05 * torch.randn(3, H, W, generator=g))
    G1, G2 = torch.rand(H * W, generator=g).cuda() * 1e-6, torch.rand(H * W, generator=g).cuda() * 1e-6  # (a mean-reduced loss's magnitudes)
    leaves = (depth, normal, depth_n, normal_n)

    def run():
        for t in leaves:
            t.grad = None
        noise, angle, valid = MV.mv_geo(depth, normal, depth_n, normal_n, ref, near, 5.0)
        ((noise * G1 + angle * G2) * valid).sum().backward()
        return [t.grad.clone() for t in leaves]

    a = [run() for _ in range(4)]
    for b in a[1:]:
        for x, y in zip(a[0], b):
            assert torch.equal(x, y), "deterministic mode: identical bits run to run"
    try:
        MV.set_deterministic(False)
        f = run()
    finally:
        MV.set_deterministic(True)
    for x, y in zip(a[0], f):
        assert torch.isfinite(x).all() and (x - y).abs().max().item() <= 2e-5 * max(y.abs().max().item(), 1e-30)


@pytest.mark.parametrize("N", [1920 * 1080, 1000, 7])
def test_fused_geo_loss_equals_the_masked_means(N):
    """utils/loss_utils.py:277-291: the fused reduction (include/gs2m_loss.h: gs2m_mv_geo_loss_*) against the PyTorch
    expressions multi_view_loss uses otherwise -- value, both masks' by-products and the gradients to noise and angle."""
    import gs2m_mvs as M
    opt = M.MultiViewParams()
    g = torch.Generator().manual_seed(N)
    noise = (torch.rand(N, generator=g) * 1.6).cuda()
    angle = (torch.rand(N, generator=g) * 1.2).cuda()
    valid = (torch.rand(N, generator=g) < 0.8).cuda()
    n0, a0 = noise.clone().requires_grad_(True), angle.clone().requires_grad_(True)
    angle_valid = valid & (a0 < opt.mv_angle_threshold * torch.pi / 180.0)
    pixel_valid = valid & (n0 < 1.0)
    geo_w = torch.where(pixel_valid, torch.exp(-n0 * opt.mv_geo_weight_decay), 0.0).detach()
    ref = opt.multi_view_geo_weight * ((geo_w * n0 * pixel_valid).sum() / pixel_valid.sum().clamp(min=1)
                                       + (geo_w * (opt.mv_angle_factor * a0) * angle_valid).sum() / angle_valid.sum().clamp(min=1))
    (ref * 3.0).backward()
    n1, a1 = noise.clone().requires_grad_(True), angle.clone().requires_grad_(True)
    got, pv, w_ncc = M.mv_geo_loss(n1, a1, valid, opt)
    (got * 3.0).backward()
    assert torch.equal(pv, pixel_valid)
    assert torch.allclose(w_ncc, torch.where(pixel_valid, torch.exp(-noise), 0.0), rtol=1e-6, atol=1e-7)
    assert abs(float(got.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach())) + 1e-12
    assert torch.allclose(n1.grad, n0.grad, rtol=1e-5, atol=1e-14) and torch.allclose(a1.grad, a0.grad, rtol=1e-5, atol=1e-14)
    assert torch.equal(M.mv_geo_loss(noise, angle, valid, opt)[0], got.detach()), "fixed-order reduction"


def test_mv_take_and_ncc_tail_match_the_framework_ops():
    """gs2m_mv_take_* / gs2m_ncc_tail_* (include/gs2m_loss.h) against the indexing and reduction expressions they replace in
    multi_view_loss (utils/loss_utils.py:293-300, 345-349): values bit for bit, gradients bit for bit (distinct indices: nothing is added up)"""
    import gs2m_mvs
    g = torch.Generator().manual_seed(11)
    H, W, n = 97, 131, 4000
    dev = "cuda"
    nm = torch.randn(3, H, W, generator=g).to(dev).requires_grad_(True)
    dm = torch.rand(1, H, W, generator=g).to(dev).requires_grad_(True)
    wm = torch.rand(H, W, generator=g).to(dev)
    idx = torch.randperm(H * W, generator=g)[:n].to(dev)
    px, ln, ld, w = gs2m_mvs._MVTake.apply(idx, nm, dm, wm)
    ix, iy = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
    grid = torch.stack([ix, iy], dim=-1).float().to(dev).reshape(-1, 2)
    assert torch.equal(px, grid[idx]) and torch.equal(ln, nm.permute(1, 2, 0).reshape(-1, 3)[idx]) and torch.equal(ld, dm.reshape(-1)[idx]) and torch.equal(w, wm.reshape(-1)[idx])
    gn, gd = torch.randn(n, 3, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    a = torch.autograd.grad((ln * gn).sum() + (ld * gd).sum(), (nm, dm))
    b = torch.autograd.grad((nm.permute(1, 2, 0).reshape(-1, 3)[idx] * gn).sum() + (dm.reshape(-1)[idx] * gd).sum(), (nm, dm))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ncc = (2.0 * torch.rand(n, 1, generator=g)).to(dev).requires_grad_(True)
    la = gs2m_mvs._NCCTail.apply(ncc, w)
    m = (ncc < 0.9).reshape(-1)
    lb = (ncc.reshape(-1) * w * m).sum() / m.sum().clamp(min=1)
    assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb))
    (ga,), (gb,) = torch.autograd.grad(3.0 * la, ncc), torch.autograd.grad(3.0 * lb, ncc)
    assert torch.allclose(ga, gb, rtol=1e-6, atol=0)
    none = torch.full((50, 1), 1.5, device=dev, requires_grad=True)  # no sample below 0.9: 0 / max(0, 1)
    assert float(gs2m_mvs._NCCTail.apply(none, torch.ones(50, device=dev))) == 0.0


def test_random_subset_on_the_device_is_an_exact_uniform_draw():
    """gs2m_mvs.random_subset on a CUDA mask (include/gs2m_loss.h: gs2m_subset_thin / gs2m_subset_remove, counter-based random numbers):
    exactly min(k, count) indices of set elements, ascending, no duplicates, every set element equally likely, the same draw for the same
    seeds; the small-count and near-k cases as in the CPU test (tests/test_losses.py)."""
    assert torch.cuda.is_available()
    import random
    import gs2m_mvs
    g = torch.Generator().manual_seed(0)
    mask = (torch.rand(200_000, generator=g) < 0.6).cuda()
    n = int(mask.sum())
    k = 30_000
    hits = torch.zeros(200_000, device="cuda")
    rng = random.Random(5)
    for _ in range(100):
        idx = gs2m_mvs.random_subset(mask, k, rng=rng)
        assert idx.dtype == torch.int64 and idx.numel() == k and bool((idx[1:] > idx[:-1]).all()) and bool(mask[idx].all())
        hits[idx] += 1
    freq = hits[mask] / 100.0
    assert abs(float(freq.mean()) - k / n) < 1e-6 and float(freq.std()) < 1.5 * (k / n * (1 - k / n) / 100.0) ** 0.5
    a = gs2m_mvs.random_subset(mask, k, rng=random.Random(7))
    b = gs2m_mvs.random_subset(mask, k, rng=random.Random(7))
    c = gs2m_mvs.random_subset(mask, k, rng=random.Random(8))
    assert torch.equal(a, b) and not torch.equal(a, c)
    few = torch.zeros(5000, dtype=torch.bool); few[::50] = True
    idx = gs2m_mvs.random_subset(few.cuda(), 3000)
    assert idx.numel() == 100 and bool(few.cuda()[idx].all())
    assert gs2m_mvs.random_subset(torch.zeros(100, dtype=torch.bool, device="cuda"), 10).numel() == 0
    near = torch.ones(3100, dtype=torch.bool, device="cuda")  # count within 4 sigma of k: everything is kept by the thinning, 100 removed
    idx = gs2m_mvs.random_subset(near, 3000)
    assert idx.numel() == 3000 and len(torch.unique(idx)) == 3000
    tiny = torch.ones(40, dtype=torch.bool, device="cuda")   # most survivors would have to go: the plain path
    assert gs2m_mvs.random_subset(tiny, 5).numel() == 5
