"""`nvdiffrast.torch.texture` (HIP, include/gs2m_texture.h) in the three modes the reference's PBR stage uses, against
the numpy restatement in oracle/texture_oracle.py and against properties any correct cube-map filter has."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _dirs(n, seed, near_edges=True):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    if near_edges:  # a third of them hugging edges / corners of the cube, where the footprint leaves the face
        k = n // 3
        d[:k] = torch.sign(d[:k]) * (1.0 - 0.02 * torch.rand(k, 3, generator=g))
        d[k:2 * k, 0] = torch.sign(d[k:2 * k, 0]) * 1.0
        d[k:2 * k, 1] = torch.sign(d[k:2 * k, 1]) * (1.0 - 0.01 * torch.rand(k, generator=g))
    return d * (0.5 + torch.rand(n, 1, generator=g))   # any length


@pytest.mark.parametrize("w,C", [(4, 3), (16, 3), (8, 1), (2, 4)])
def test_cube_linear_matches_oracle(w, C):
    assert torch.cuda.is_available()
    import nvdiffrast.torch as dr
    from oracle import texture_oracle as O
    g = torch.Generator().manual_seed(w * 10 + C)
    tex = torch.rand(1, 6, w, w, C, generator=g)
    d = _dirs(600, w)
    got = dr.texture(tex.cuda(), d.view(1, 20, 30, 3).cuda(), filter_mode="linear", boundary_mode="cube").cpu().view(-1, C)
    want = O.cube_sample([tex[0].numpy()], d.numpy())
    assert np.abs(got.numpy() - want).max() < 3e-5   # fp32 texel-space coordinates: (u * w - 0.5) loses ~w ulp


def test_cube_mip_linear_matches_oracle():
    assert torch.cuda.is_available()
    import nvdiffrast.torch as dr
    from oracle import texture_oracle as O
    g = torch.Generator().manual_seed(5)
    levels = [torch.rand(1, 6, w, w, 3, generator=g) for w in (32, 16, 8, 4)]
    d = _dirs(900, 9)
    bias = torch.rand(900, generator=g) * 4.5 - 0.7         # below 0 and above the last level included
    bias[:50] = torch.tensor([0.0, 1.0, 2.0, 3.0, 2.999]).repeat(10)
    got = dr.texture(levels[0].cuda(), d.view(1, 30, 30, 3).cuda(), mip=[l.cuda() for l in levels[1:]],
                     mip_level_bias=bias.view(1, 30, 30).cuda(), filter_mode="linear-mipmap-linear", boundary_mode="cube").cpu().view(-1, 3)
    want = O.cube_sample([l[0].numpy() for l in levels], d.numpy(), bias.numpy())
    assert np.abs(got.numpy() - want).max() < 3e-5


def test_tex2d_clamp_matches_oracle():
    assert torch.cuda.is_available()
    import nvdiffrast.torch as dr
    from oracle import texture_oracle as O
    g = torch.Generator().manual_seed(6)
    tex = torch.rand(1, 24, 40, 2, generator=g)
    uv = torch.rand(800, 2, generator=g) * 1.2 - 0.1
    uv[:8] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.5 / 40, 0.5 / 24], [1 - 0.5 / 40, 0.3], [0.3, 1 - 0.5 / 24], [0.5, 0.5], [-1.0, 2.0], [1.5 / 40, 1.5 / 24]])
    got = dr.texture(tex.cuda(), uv.view(1, 20, 40, 2).cuda(), filter_mode="linear", boundary_mode="clamp").cpu().view(-1, 2)
    want = O.tex2d_clamp_sample(tex[0].numpy(), uv.numpy())
    assert np.abs(got.numpy() - want).max() < 3e-5


def test_cube_filter_properties():
    """Oracle-free: constants stay constant, a texel-centre direction returns that texel, the filter is continuous
    across every edge and around every corner, non-finite directions give zero."""
    assert torch.cuda.is_available()
    import nvdiffrast.torch as dr
    dev = "cuda"
    w = 8
    g = torch.Generator().manual_seed(7)
    tex = torch.rand(1, 6, w, w, 3, generator=g).to(dev)
    look = lambda d: dr.texture(tex, d.view(1, 1, -1, 3).contiguous(), filter_mode="linear", boundary_mode="cube").view(-1, 3)
    d = _dirs(3000, 1).to(dev)
    const = dr.texture(torch.full_like(tex, 0.37), d.view(1, 30, 100, 3), filter_mode="linear", boundary_mode="cube")
    assert (const - 0.37).abs().max().item() < 1e-6
    # texel centres (pbr/light.py:13-26 table)
    c = (2 * (torch.arange(w, dtype=torch.float32) + 0.5) / w - 1).to(dev)
    Y, X = torch.meshgrid(c, c, indexing="ij")
    one = torch.ones_like(X)
    faces = [(one, -Y, -X), (-one, -Y, X), (X, one, Y), (X, -one, -Y), (X, -Y, one), (-X, -Y, -one)]
    for f, comp in enumerate(faces):
        got = look(torch.stack(comp, dim=-1).reshape(-1, 3)).view(w, w, 3)
        assert (got - tex[0, f]).abs().max().item() < 1e-5, f
    # continuity: points straddling the edges / corners of the cube
    gen = torch.Generator().manual_seed(8)
    p = torch.rand(4000, 3, generator=gen) * 2 - 1
    p[:, 0] = 1.0
    p[:, 1] = torch.sign(p[:, 1]) * 1.0                       # on the edge between face +-x ... and face +-y
    p[:1000, 2] = torch.sign(p[:1000, 2]) * 1.0               # the first thousand on a corner
    p = p[:, torch.randperm(3, generator=gen)] * torch.sign(torch.randn(1, 3, generator=gen))
    eps = 1e-4 * torch.randn(4000, 3, generator=gen)
    a, b = look((p + eps).to(dev)), look((p - eps).to(dev))
    assert (a - b).abs().max().item() < 5e-3 * w / 8, (a - b).abs().max().item()   # Lipschitz: |grad| <= w * range
    bad = torch.tensor([[float("nan"), 0.0, 1.0], [0.0, 0.0, 0.0], [float("inf"), float("inf"), 1.0]], device=dev)   # u or v not finite
    assert look(bad).abs().max().item() == 0.0


@pytest.mark.parametrize("mode", ["cube", "cube_mip", "2d"])
def test_texture_backward_is_the_adjoint(mode):
    """The lookup is linear in the texture(s): <dy, T tex> == <T^t dy, tex> for random tex, dy (fp32 accumulation)."""
    assert torch.cuda.is_available()
    import nvdiffrast.torch as dr
    dev = "cuda"
    g = torch.Generator().manual_seed(11)
    n = 5000
    if mode == "2d":
        texs = [torch.rand(1, 16, 32, 2, generator=g).to(dev).requires_grad_(True)]
        uv = (torch.rand(1, 50, 100, 2, generator=g) * 1.2 - 0.1).to(dev)
        out = dr.texture(texs[0], uv, filter_mode="linear", boundary_mode="clamp")
    else:
        ws = (16, 8, 4) if mode == "cube_mip" else (16,)
        texs = [torch.rand(1, 6, w, w, 3, generator=g).to(dev).requires_grad_(True) for w in ws]
        d = _dirs(n, 12).view(1, 50, 100, 3).to(dev)
        if mode == "cube_mip":
            bias = (torch.rand(1, 50, 100, generator=g) * 3.0 - 0.5).to(dev)
            out = dr.texture(texs[0], d, mip=texs[1:], mip_level_bias=bias, filter_mode="linear-mipmap-linear", boundary_mode="cube")
        else:
            out = dr.texture(texs[0], d, filter_mode="linear", boundary_mode="cube")
    dy = torch.randn(out.shape, generator=g).to(dev)
    lhs = (out.double() * dy.double()).sum().item()
    grads = torch.autograd.grad(out, texs, dy, retain_graph=True)
    rhs = sum((gt.double() * t.detach().double()).sum().item() for gt, t in zip(grads, texs))
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    # and the gradient of a single pixel touches at most 4 texels per level, with weights summing to the pixel's share
    one = torch.zeros_like(dy)
    one.view(-1, dy.shape[-1])[123] = 1.0
    g1 = torch.autograd.grad(out, texs, one)
    tot = sum(x.sum().item() for x in g1)
    assert abs(tot - dy.shape[-1]) < 1e-5 and all((x != 0).sum().item() <= 4 * dy.shape[-1] for x in g1)


def test_texture_refuses_what_it_does_not_implement():
    assert torch.cuda.is_available()
    import nvdiffrast.torch as dr
    tex = torch.rand(1, 6, 4, 4, 3, device="cuda")
    d = torch.randn(1, 2, 2, 3, device="cuda")
    with pytest.raises(NotImplementedError):
        dr.texture(tex, d, filter_mode="nearest", boundary_mode="cube")
    with pytest.raises(NotImplementedError):
        dr.texture(tex, d, uv_da=torch.zeros(1, 2, 2, 6, device="cuda"), boundary_mode="cube")
    with pytest.raises(NotImplementedError):
        dr.texture(tex, d.requires_grad_(True), filter_mode="linear", boundary_mode="cube")
    with pytest.raises(NotImplementedError):
        dr.texture(torch.rand(1, 4, 4, 3, device="cuda"), torch.rand(1, 2, 2, 2, device="cuda"), filter_mode="linear", boundary_mode="wrap")
    with pytest.raises(RuntimeError, match="no CPU path"):
        dr.texture(tex.cpu(), d.detach().cpu(), filter_mode="linear", boundary_mode="cube")
