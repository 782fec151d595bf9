"""The deferred PBR stage (pbr/ package on the HIP texture / prefilter operators): mip construction, the smoothed
cubemap_mip backward, shading values against a numpy evaluation through the oracles, gradients to light and albedo."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_cubemap_mip_forward_and_smoothed_backward():
    assert torch.cuda.is_available()
    from pbr.light import cubemap_mip, _texel_center_dirs
    from oracle import texture_oracle as O
    g = torch.Generator().manual_seed(0)
    x = torch.rand(6, 8, 8, 3, generator=g).cuda().requires_grad_(True)
    y = cubemap_mip.apply(x)
    want = x.detach().view(6, 4, 2, 4, 2, 3).mean(dim=(2, 4))
    assert torch.allclose(y, want, atol=1e-6)
    G = torch.randn(6, 4, 4, 3, generator=g).cuda()
    (gx,) = torch.autograd.grad(y, x, G)
    dirs = _texel_center_dirs(8, "cuda").view(-1, 3).cpu().numpy()
    ref = O.cube_sample([(G * 0.25).cpu().numpy()], dirs).reshape(6, 8, 8, 3)   # pbr/light.py:36-48
    assert np.abs(gx.cpu().numpy() - ref).max() < 1e-5


def _light(res=64, seed=1):
    from pbr import CubemapLight
    torch.manual_seed(seed)
    return CubemapLight(base_res=res)


def test_build_mips_and_get_mip():
    assert torch.cuda.is_available()
    light = _light(64)
    light.build_mips()
    assert [tuple(s.shape) for s in light.specular] == [(6, 64, 64, 3), (6, 32, 32, 3), (6, 16, 16, 3)]
    assert tuple(light.diffuse.shape) == (6, 16, 16, 3)
    r = torch.tensor([0.0, 0.04, 0.27, 0.4999, 0.5, 0.75, 1.0], device="cuda")
    n = 3
    want = torch.tensor([0.0, 0.0, 0.5 * (n - 2), (0.4999 - 0.04) / 0.46 * (n - 2), n - 2.0, n - 1.5, n - 1.0], device="cuda")
    assert torch.allclose(light.get_mip(r), want, atol=1e-5)
    # a constant environment: every prefiltered level equals that constant, the irradiance the constant times the
    # cosine-lobe weight sum (~1.05 .. 1.12 with the texel-area proxy at 16^2)
    with torch.no_grad():
        light.base.fill_(0.5)
    light.build_mips()
    for s in light.specular:
        assert (s - 0.5).abs().max().item() < 1e-5
    assert 0.5 * 1.0 < light.diffuse.min().item() and light.diffuse.max().item() < 0.5 * 1.15


def test_pbr_shading_matches_numpy_evaluation_and_has_gradients():
    assert torch.cuda.is_available()
    from pbr import get_brdf_lut, pbr_shading
    from oracle import texture_oracle as O
    light = _light(64, seed=2)
    light.build_mips()
    lut = get_brdf_lut().cuda()
    assert tuple(lut.shape) == (1, 256, 256, 2) and 0.0 <= lut.min().item() and lut.max().item() <= 1.0 + 1e-3
    H, W = 12, 20
    g = torch.Generator().manual_seed(3)
    n = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1).cuda()
    v = torch.nn.functional.normalize(n.cpu() + 0.8 * torch.randn(H, W, 3, generator=g), dim=-1).cuda()
    albedo = torch.rand(H, W, 3, generator=g).cuda().requires_grad_(True)
    rough = (0.04 + 0.96 * torch.rand(H, W, 1, generator=g)).cuda()
    metal = torch.rand(H, W, 1, generator=g).cuda().requires_grad_(True)
    pkg = pbr_shading(light, n, v, albedo, rough, metallic=metal, occlusion=torch.ones_like(rough), irradiance=torch.zeros_like(rough), brdf_lut=lut)
    # the same in numpy through the texture oracle
    nn_, vv = n.cpu().double().numpy().reshape(-1, 3), v.cpu().double().numpy().reshape(-1, 3)
    al, ro, me = albedo.detach().cpu().double().numpy().reshape(-1, 3), rough.cpu().double().numpy().reshape(-1, 1), metal.detach().cpu().double().numpy().reshape(-1, 1)
    ndv = (nn_ * vv).sum(-1, keepdims=True)
    refl = 2.0 * np.clip(ndv, 0.0, None) * nn_ - vv
    diff = O.cube_sample([light.diffuse.detach().cpu().numpy()], nn_) * al
    fg = O.tex2d_clamp_sample(lut[0].cpu().numpy(), np.concatenate([np.clip(ndv, 1e-4, 1.0), ro], axis=1))
    mip = light.get_mip(rough).cpu().numpy().reshape(-1)
    spec = O.cube_sample([s.detach().cpu().numpy() for s in light.specular], refl, mip)
    F0 = (1.0 - me) * 0.04 + al * me
    want = np.clip(diff + spec * (F0 * fg[:, 0:1] + fg[:, 1:2]), 0.0, 1.0)
    assert np.abs(pkg["render_rgb"].detach().cpu().numpy().reshape(-1, 3) - want).max() < 1e-4
    # gradients reach the light, the albedo and the metallic map; the light's is checked along a random direction
    Gw = torch.rand(H, W, 3, generator=g).cuda()
    loss = (pkg["render_rgb"] * Gw).sum()
    g_base, g_alb, g_met = torch.autograd.grad(loss, [light.base, albedo, metal])
    assert g_base.abs().sum().item() > 0 and g_alb.abs().sum().item() > 0 and g_met.abs().sum().item() > 0
    d = torch.randn(light.base.shape, generator=g).cuda()

    def f(eps):
        with torch.no_grad():
            light.base.add_(eps * d)
        light.build_mips()
        out = pbr_shading(light, n, v, albedo.detach(), rough, metallic=metal.detach(), occlusion=torch.ones_like(rough),
                          irradiance=torch.zeros_like(rough), brdf_lut=lut)["render_rgb"]
        with torch.no_grad():
            light.base.sub_(eps * d)
        return (out.double() * Gw.double()).sum().item()

    fd = (f(1e-2) - f(-1e-2)) / 2e-2
    an = (g_base.double() * d.double()).sum().item()
    assert abs(fd - an) < 2e-2 * max(1.0, abs(an)), (fd, an)


def test_environment_light_is_learnable_through_the_whole_stack():
    """Fit the light to reproduce a target shading of a fixed G-buffer: the loss must fall (gradients through the texture
    lookups, both prefilters and the smoothed mip chain, Adam on the cube map)."""
    assert torch.cuda.is_available()
    from pbr import get_brdf_lut, pbr_shading
    import gs2m_optim
    H, W = 64, 96
    g = torch.Generator().manual_seed(5)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    n = torch.nn.functional.normalize(torch.stack([xx, yy, 1.1 - xx * xx - yy * yy], dim=-1), dim=-1).cuda()
    v = torch.nn.functional.normalize(torch.stack([0.2 * xx, 0.2 * yy, torch.ones_like(xx)], dim=-1), dim=-1).cuda()
    albedo, rough = torch.rand(H, W, 3, generator=g).cuda(), (0.1 + 0.8 * torch.rand(H, W, 1, generator=g)).cuda()
    lut = get_brdf_lut().cuda()
    args = dict(occlusion=torch.ones_like(rough), irradiance=torch.zeros_like(rough), brdf_lut=lut)
    target_light = _light(64, seed=7)
    target_light.build_mips()
    with torch.no_grad():
        target = pbr_shading(target_light, n, v, albedo, rough, **args)["render_rgb"]
    light = _light(64, seed=8)
    opt = gs2m_optim.Adam(light.parameters(), lr=0.02)
    losses = []
    for it in range(60):
        light.build_mips()
        loss = (pbr_shading(light, n, v, albedo, rough, **args)["render_rgb"] - target).abs().mean()
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        light.clamp_(min=0.0)
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


@pytest.mark.parametrize("with_metallic", [True, False])
def test_fused_shading_equals_the_op_by_op_form(with_metallic):
    """pbr_shading_fused (one kernel each way) against pbr_shading (PyTorch ops around dr.texture): every output, and the
    gradients to albedo, metallic and the light's base map through build_mips."""
    assert torch.cuda.is_available()
    from pbr import get_brdf_lut, pbr_shading, pbr_shading_fused
    H, W = 45, 70
    g = torch.Generator().manual_seed(21)
    n = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)
    n[:3] = 0.0                                                # background rows: zero normals
    n = n.cuda()
    v = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1).cuda()
    rough = (0.04 + 0.96 * torch.rand(H, W, 1, generator=g)).cuda()
    lut = get_brdf_lut().cuda()
    Gw = torch.randn(H, W, 3, generator=g).cuda()
    res = {}
    for fused in (False, True):
        light = _light(64, seed=22)
        with torch.no_grad():
            light.base.mul_(2.5)                               # push part of the image over 1: the clamp must gate the gradient
        albedo = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(23)).cuda().requires_grad_(True)
        metal = torch.rand(H, W, 1, generator=torch.Generator().manual_seed(24)).cuda().requires_grad_(True)
        light.build_mips()
        if fused:
            pkg = pbr_shading_fused(light, n, v, albedo, rough, metallic=metal if with_metallic else None, brdf_lut=lut)
        else:
            pkg = pbr_shading(light, n, v, albedo, rough, metallic=metal if with_metallic else None, occlusion=torch.ones_like(rough),
                              irradiance=torch.zeros_like(rough), brdf_lut=lut)
        (pkg["render_rgb"] * Gw).sum().backward()
        res[fused] = (pkg, albedo.grad, metal.grad, light.base.grad)
    for k in ("render_rgb", "diffuse_rgb", "specular_rgb", "diffuse_light"):
        assert (res[True][0][k].reshape(H, W, 3) - res[False][0][k].reshape(H, W, 3)).abs().max().item() < 2e-5, k
    assert 0.02 < (res[False][0]["render_rgb"] >= 1.0).float().mean().item() < 0.98
    assert (res[True][1] - res[False][1]).abs().max().item() < 1e-4 * max(1.0, res[False][1].abs().max().item())
    if with_metallic:
        assert (res[True][2] - res[False][2]).abs().max().item() < 1e-4 * max(1.0, res[False][2].abs().max().item())
    else:
        assert res[True][2] is None and res[False][2] is None
    gb0, gb1 = res[False][3], res[True][3]
    assert (gb1 - gb0).abs().max().item() < 2e-4 * max(1.0, gb0.abs().max().item())


@pytest.mark.parametrize("with_metallic", [True, False])
def test_pbr_render_fused_inputs_equal_the_op_by_op_preparation(with_metallic):
    """pbr_render end to end (pbr/__init__.py:9-56): the fused preparation of the shading inputs + fused shading against the
    PyTorch preparation + op-by-op shading, from the same planar G-buffer maps: image, returned maps and the gradients that
    reach the albedo map (through its clamp), a learnt metallic map and the light."""
    import types
    from pbr import get_brdf_lut, pbr_render
    H, W = 37, 52
    g = torch.Generator().manual_seed(5)
    maps = {"normal_map": torch.randn(3, H, W, generator=g) * 0.7, "albedo_map": torch.rand(3, H, W, generator=g) * 1.4 - 0.2,
            "roughness_map": torch.rand(1, H, W, generator=g) * 1.2 - 0.1, "alpha_map": torch.rand(1, H, W, generator=g),
            "metallic_map": torch.rand(1, H, W, generator=g)}
    maps["normal_map"][:, :2] = 0.0   # background rows: zero normals stay zero
    rays = torch.nn.functional.normalize(torch.randn(H * W, 3, generator=g), dim=-1).cuda()
    cam = types.SimpleNamespace(image_height=H, image_width=W, world_view_transform=torch.eye(4).cuda())
    Gw = torch.randn(H, W, 3, generator=g).cuda()
    res = {}
    for fused in (False, True):
        scene = types.SimpleNamespace(cubemap=_light(64, seed=9), brdf_lut=get_brdf_lut().cuda())
        pkg_in = {k: v.clone().cuda().requires_grad_(k in ("albedo_map", "metallic_map", "roughness_map")) for k, v in maps.items()}
        pkg = pbr_render(scene, cam, rays, pkg_in, metallic=with_metallic, fused=fused)
        (pkg["render_rgb"].reshape(H, W, 3) * Gw).sum().backward()
        res[fused] = (pkg, pkg_in["albedo_map"].grad, pkg_in["metallic_map"].grad, scene.cubemap.base.grad, pkg_in["roughness_map"].grad)
    a, b = res[False], res[True]
    assert (a[0]["render_rgb"].reshape(H, W, 3) - b[0]["render_rgb"].reshape(H, W, 3)).abs().max().item() < 3e-5
    for k in ("roughness_map", "metallic_map"):
        assert a[0][k].shape == b[0][k].shape == (1, H, W) and torch.allclose(a[0][k], b[0][k], atol=1e-7), k
    assert torch.allclose(b[1], a[1], rtol=1e-4, atol=1e-6) and float((a[1] == 0).float().mean()) > 0.1, "the clamp gates the albedo gradient"
    if with_metallic:
        assert torch.allclose(b[2], a[2], rtol=1e-4, atol=1e-6)
    else:
        assert a[2] is None and b[2] is None
    assert b[4] is None and a[4] is None, "the roughness is detached"
    assert (b[3] - a[3]).abs().max().item() < 2e-4 * max(1.0, a[3].abs().max().item())
