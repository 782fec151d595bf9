"""Fused render() pre/post-processing kernels (csrc/render_ops.hip) against the PyTorch
formulation of the reference (gaussian_renderer/__init__.py:83-96, 126-141; scene/gaussian_model.py:146-160), which
is the fp32 reference here: forward values within 1e-6 (relative to the channel magnitude), gradients within 1e-5."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def _close(name, a, b, tol):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{name}: max error {err:.3e} (scale {scale:.3e})"


def _scene(P, seed, dev):
    import gs2m_synth as S
    from gs2m_scene import GaussianParams, Camera
    cam0 = S.make_camera(96, 64)
    g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam0, seed=seed, scale_hi=0.08).items()}
    gen = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.rand(*s, generator=gen) * 0.8 + 0.1).to(dev)
    pc = GaussianParams.from_activated(g["means3D"], g["shs"], g["scales"], g["rotations"] * 1.7, g["opacities"].clamp(0.02, 0.98),
                                       r(P, 3), r(P, 1), r(P, 1))
    for t in pc.parameters():
        t.requires_grad_(True)
    return pc, Camera(cam0, dev)


def _torch_features(pc, cam, z_depth, blend_metallic):  # GR:83-96
    means3D = pc.get_xyz
    normals = pc.get_normals(cam.camera_center)
    cam_normals = normals @ cam.world_view_transform[:3, :3]
    cam_points = means3D @ cam.world_view_transform[:3, :3] + cam.world_view_transform[3, :3]
    f = torch.zeros((means3D.shape[0], 10), dtype=torch.float32, device=means3D.device)
    f[:, 0] = 1.0
    f[:, 1] = cam_points[:, 2] if z_depth else (cam_normals * cam_points).sum(dim=-1).abs()
    f[:, 2:5] = normals
    f[:, 5:8] = pc.get_albedo
    f[:, 8:9] = pc.get_roughness
    if blend_metallic:
        f[:, 9:10] = pc.get_metallic
    return f


@pytest.mark.parametrize("z_depth,blend_metallic", [(False, False), (False, True), (True, True)])
def test_pack_features_matches_torch(z_depth, blend_metallic):
    assert torch.cuda.is_available()
    import gs2m_render_ops as R
    dev = "cuda"
    pc, cam = _scene(5000, 3, dev)
    G = torch.randn(5000, 10, generator=torch.Generator().manual_seed(1)).to(dev)
    ref = _torch_features(pc, cam, z_depth, blend_metallic)
    (ref * G).sum().backward()
    gref = [None if t.grad is None else t.grad.clone() for t in pc.parameters()]
    for t in pc.parameters():
        t.grad = None
    out = R.pack_features(pc.get_xyz, pc.get_scaling, pc.get_rotation, pc.get_albedo, pc.get_roughness, pc.get_metallic,
                          cam.camera_center, cam.world_view_transform, z_depth=z_depth, blend_metallic=blend_metallic)
    assert torch.equal(out[:, 0], ref[:, 0]) and torch.equal(out[:, 9] == 0, ref[:, 9] == 0)
    assert torch.equal(torch.sign(out[:, 2:5]), torch.sign(ref[:, 2:5])), "axis pick / flip must agree exactly"
    _close("features", out, ref, 1e-6)
    (out * G).sum().backward()
    names = ["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic"]
    for n, t, gr in zip(names, pc.parameters(), gref):
        if gr is None or t.grad is None:
            assert (gr is None or float(gr.abs().max()) == 0.0) and (t.grad is None or float(t.grad.abs().max()) == 0.0), n
            continue
        _close("grad " + n, t.grad, gr, 1e-5)


@pytest.mark.parametrize("z_depth", [False, True])
def test_gbuffer_post_matches_torch(z_depth):
    assert torch.cuda.is_available()
    import gs2m_render_ops as R
    import gs2m_synth as S
    from gs2m_scene import Camera
    dev = "cuda"
    H, W = 37, 53
    cam = Camera(S.make_camera(W, H), dev)
    gen = torch.Generator().manual_seed(5)
    buf = torch.randn(10, H, W, generator=gen).to(dev)
    buf[2:5, :5] = 0.0            # background rows: normal mask off
    buf[3, 7, 11] = 0.0           # a single zero channel also switches the mask off
    buf[1] = buf[1].abs() + 0.5
    buf.requires_grad_(True)
    Gl, Gd = torch.randn(3, H, W, generator=gen).to(dev), torch.randn(1, H, W, generator=gen).to(dev)

    def torch_post(b):
        normal_map = b[2:5]
        mask = (normal_map != 0).all(0, keepdim=True)
        ln = normal_map.permute(1, 2, 0).view(-1, 3) @ cam.world_view_transform[:3, :3]
        lnm = ln.reshape(H, W, 3).permute(2, 0, 1)
        depth = b[1:2]
        if not z_depth:
            denoms = torch.sum(ln * cam.get_rays().view(-1, 3), dim=-1).view(1, H, W)
            depth = b[1:2] / -(denoms + 1e-8)
        return mask, lnm, depth

    m0, l0, d0 = torch_post(buf)
    ((l0 * Gl).sum() + (d0 * Gd).sum()).backward()
    g0 = buf.grad.clone()
    buf.grad = None
    rays = None if z_depth else cam.get_rays().view(-1, 3)
    m1, l1, d1 = R.gbuffer_post(buf, rays, cam.world_view_transform, z_depth=z_depth)
    assert m1.dtype == torch.bool and torch.equal(m1, m0)
    _close("local_normal_map", l1, l0, 1e-6)
    _close("depth_map", d1, d0, 1e-5)
    ((l1 * Gl).sum() + (d1 * Gd).sum()).backward()
    _close("grad buffer", buf.grad, g0, 1e-5)


@pytest.mark.parametrize("z_depth", [False, True])
@pytest.mark.parametrize("used", ["all", "some", "post_only"])
def test_gbuffer_maps_matches_slices(z_depth, used):
    """gbuffer_maps = the channel slices + gbuffer_post as one autograd node: same maps, and the same dL/dbuffer
    whichever subset of the maps the loss touches (the untouched ones arrive as NULL gradients)."""
    assert torch.cuda.is_available()
    import gs2m_render_ops as R
    import gs2m_synth as S
    from gs2m_scene import Camera
    dev = "cuda"
    H, W = 41, 67
    cam = Camera(S.make_camera(W, H), dev)
    gen = torch.Generator().manual_seed(11)
    buf = torch.randn(10, H, W, generator=gen).to(dev)
    buf[2:5, :4] = 0.0
    buf[1] = buf[1].abs() + 0.5
    buf.requires_grad_(True)
    G = [torch.randn(c, H, W, generator=gen).to(dev) for c in (1, 1, 3, 3, 1, 1, 3, 1)]
    rays = None if z_depth else cam.get_rays().view(-1, 3)
    pick = {"all": range(8), "some": (0, 2, 4, 7), "post_only": (6, 7)}[used]

    def loss(maps):
        return sum((maps[k] * G[k]).sum() for k in pick)

    m0, l0, d0 = R.gbuffer_post(buf, rays, cam.world_view_transform, z_depth=z_depth)
    ref = [buf[0:1], buf[1:2], buf[2:5], buf[5:8], buf[8:9], buf[9:10], l0, d0]
    loss(ref).backward()
    g0 = buf.grad.clone()
    buf.grad = None
    out = R.gbuffer_maps(buf, rays, cam.world_view_transform, z_depth=z_depth)
    got = list(out[:6]) + [out[7], out[8]]
    assert torch.equal(out[6], m0)
    for k in range(8):
        assert torch.equal(got[k], ref[k])
    for k in range(6):
        assert got[k].data_ptr() == ref[k].data_ptr()  # slices stay views of the rasterizer's buffer
    loss(got).backward()
    _close("grad buffer", buf.grad, g0, 1e-6)


@pytest.mark.parametrize("P", [1, 255, 4099])
def test_activate_matches_model_getters(P):
    """The six GaussianModel getters (exp, F.normalize, sigmoid x4; GM:113-144) as one launch: values and gradients."""
    assert torch.cuda.is_available()
    import torch.nn.functional as F
    import gs2m_render_ops as R
    gen = torch.Generator().manual_seed(P)
    shapes = [(P, 3), (P, 4), (P, 1), (P, 3), (P, 1), (P, 1)]
    raw = [(torch.randn(s, generator=gen) * 2.0).cuda().requires_grad_(True) for s in shapes]
    with torch.no_grad():
        raw[1][0] = 0.0      # a zero quaternion: below F.normalize's eps clamp
    G = [torch.randn(s, generator=gen).cuda() for s in shapes]
    fns = [torch.exp, F.normalize, torch.sigmoid, torch.sigmoid, torch.sigmoid, torch.sigmoid]
    ref = [f(t) for f, t in zip(fns, raw)]
    sum((a * g).sum() for a, g in zip(ref[:5], G)).backward()      # metallic gets no gradient
    g_ref = [None if t.grad is None else t.grad.clone() for t in raw]
    for t in raw:
        t.grad = None
    got = R.activate(*raw)
    for k, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b) if k != 1 else (a - b).abs().max().item() <= 1.2e-7, k   # exp / sigmoid bit-equal, normalize 1 ulp
    sum((a * g).sum() for a, g in zip(got[:5], G)).backward()
    for k, (t, g) in enumerate(zip(raw, g_ref)):
        assert (t.grad is None) == (g is None), k
        if g is not None:
            assert (t.grad - g).abs().max().item() <= 2e-6 * max(1.0, g.abs().max().item()), k


@pytest.mark.parametrize("material_stage,blend_metallic", [(True, False), (True, True), (False, False)])
def test_render_fused_equals_unfused(material_stage, blend_metallic):
    """render() end to end: fused pre/post-processing against the reference's PyTorch formulation around the same
    rasterizer -- every output map and every parameter gradient."""
    assert torch.cuda.is_available()
    from gs2m_scene import PipelineParams
    from gaussian_renderer import render
    dev = "cuda"
    pc, cam = _scene(3000, 9, dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gen = torch.Generator().manual_seed(2)
    H, W = cam.image_height, cam.image_width
    keys = ["render", "alpha_map", "depth_map", "normal_map", "albedo_map", "roughness_map", "metallic_map", "local_normal_map", "sobel_map"]
    res = {}
    for fused in (False, True):
        pipe = PipelineParams()
        pipe.fused_render_ops = fused
        for t in pc.parameters():
            t.grad = None
        out = render(cam, pc, pipe, bg, geometry_stage=not material_stage, material_stage=material_stage, blend_metallic=blend_metallic,
                     sobel_normal=True)
        g2 = torch.Generator().manual_seed(4)
        loss = sum((out[k] * torch.rand(out[k].shape, generator=g2).to(dev)).sum() for k in keys)
        loss.backward()
        res[fused] = ({k: out[k].detach().clone() for k in keys + ["normal_mask"]},
                      [None if t.grad is None else t.grad.clone() for t in pc.parameters()])
    assert torch.equal(res[True][0]["normal_mask"], res[False][0]["normal_mask"])
    for k in keys:
        _close(k, res[True][0][k], res[False][0][k], 2e-5)
    names = ["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic"]
    for n, a, b in zip(names, res[True][1], res[False][1]):
        assert (a is None) == (b is None), n
        if a is not None:
            _close("grad " + n, a, b, 1e-3)  # the depth division amplifies fp32 rounding; 1e-3 is the gradient bar of the path


@pytest.mark.parametrize("H,W", [(45, 61), (16, 16), (33, 17), (3, 3), (64, 130)])
def test_sobel_normal_matches_torch(H, W):
    """depth -> world points -> cross product normals -> alpha blend with the background: fused kernel (forward and the
    gather-form backward, 16x16 tiles with a halo: sizes on, below and across the tile edges) against utils/normal_utils.py
    as restated in gs2m_scene.normal_from_depth_image."""
    assert torch.cuda.is_available()
    import gs2m_render_ops as R
    import gs2m_synth as S
    from gs2m_scene import Camera
    from gaussian_renderer import render_normal_from_depth_map
    dev = "cuda"
    cam = Camera(S.look_at_camera(W, H, (1.0, -0.7, 0.5), (0.2, 0.1, 6.0)), dev)
    gen = torch.Generator().manual_seed(8)
    depth = (torch.rand(H, W, generator=gen) * 3.0 + 2.0).to(dev).requires_grad_(True)
    alpha = torch.rand(H, W, generator=gen).to(dev).requires_grad_(True)
    bg = torch.tensor([0.3, 0.1, 0.7], device=dev)
    G = torch.randn(3, H, W, generator=gen).to(dev)
    ref = render_normal_from_depth_map(cam, depth, bg, alpha, fused=False)
    (ref * G).sum().backward()
    gd, ga = depth.grad.clone(), alpha.grad.clone()
    depth.grad = None; alpha.grad = None
    out = render_normal_from_depth_map(cam, depth, bg, alpha, fused=True)
    _close("sobel_map", out, ref, 2e-5)
    (out * G).sum().backward()
    _close("grad depth", depth.grad, gd, 1e-4)
    _close("grad alpha", alpha.grad, ga, 2e-5)
    # the border carries no normal: output = background * (1 - alpha)
    assert torch.allclose(out[:, 0, :], bg[:, None] * (1 - alpha[0, :])[None], atol=1e-7)


@pytest.mark.parametrize("P", [1, 255, 256, 257, 3001])
def test_split_sh_equals_concatenated(P):
    """SH coefficients handed over as (DC, rest) -- the reference model's two parameters -- give bit-identical images and
    the gradients of the concatenated call, sliced (block boundaries of the LDS staging: 256 Gaussians)."""
    assert torch.cuda.is_available()
    import helpers as Hh
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = "cuda"
    sc = Hh.make_scene(P, 96, 64, seed=31 + P, fc=9, scale_hi=0.1)
    g = {k: v.to(dev) for k, v in sc["g"].items()}
    st = Hh.settings_for(sc, dev)
    Gc, Gb = sc["Gc"].to(dev), sc["Gb"].to(dev)
    res = []
    for split in (False, True):
        leaves = {k: v.clone().requires_grad_(True) for k, v in g.items() if k != "shs"}
        if split:
            dc = g["shs"][:, :1].contiguous().requires_grad_(True)
            rest = g["shs"][:, 1:].contiguous().requires_grad_(True)
            kw = dict(shs=dc, shs_rest=rest)
        else:
            sh = g["shs"].clone().requires_grad_(True)
            kw = dict(shs=sh)
        m2 = torch.zeros(P, 4, device=dev, requires_grad=True)
        color, radii, observe, buffer = GaussianRasterizer(st)(leaves["means3D"], m2, leaves["opacities"], scales=leaves["scales"],
                                                               rotations=leaves["rotations"], features=leaves["features"], **kw)
        ((color * Gc).sum() + (buffer * Gb).sum()).backward()
        gsh = torch.cat((dc.grad, rest.grad), dim=1) if split else sh.grad
        res.append((color.detach(), buffer.detach(), gsh, leaves["means3D"].grad, leaves["scales"].grad, m2.grad))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


def test_render_without_shading_gives_the_same_geometry_and_gradients():
    """render(..., shade=False) -- what the multi-view term asks of its neighbour view (gs2m_mvs.multi_view_loss) -- against the
    ordinary call on a loss that reads the depth and normal maps only: the maps and every parameter gradient bit for bit (the SH
    gradients: zeros in one case, absent in the other), the image black."""
    import gaussian_renderer
    import gs2m_scene
    import helpers as Hh
    sc = Hh.make_scene(15_000, 320, 200, seed=9, fc=10)
    cam = gs2m_scene.Camera(sc["cam"], "cuda")
    Gd, Gn = sc["Gb"][1:2].cuda(), sc["Gb"][2:5].cuda()

    def run(shade):
        pc = Hh.model_from_scene(sc, "cuda", requires_grad=True)
        out = gaussian_renderer.render(cam, pc, gs2m_scene.PipelineParams(), sc["bg"].cuda(), geometry_stage=True, material_stage=False, shade=shade)
        ((out["depth_map"] * Gd).sum() + (out["normal_map"] * Gn).sum()).backward()
        return out, pc

    a, pa = run(True)
    b, pb = run(False)
    assert float(b["render"].abs().max()) == 0.0 and float(a["render"].abs().max()) > 0.0
    for k in ("depth_map", "normal_map", "alpha_map", "radii", "observe"):
        assert torch.equal(a[k], b[k]), k
    for name in ("_xyz", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(pa, name).grad, getattr(pb, name).grad), name
    for name in ("_features_dc", "_features_rest"):
        ga, gb = getattr(pa, name).grad, getattr(pb, name).grad
        assert gb is None and (ga is None or float(ga.abs().max()) == 0.0), name
    assert torch.equal(a["viewspace_points"].grad, b["viewspace_points"].grad)
