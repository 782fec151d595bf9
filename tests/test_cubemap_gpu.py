"""render_utils.diffuse_cubemap / specular_cubemap (HIP, include/gs2m_cubemap.h) against dense-matrix restatements of
the reference's kernels (oracle/cubemap_oracle.py): forward = W x, backward = W^T g."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize("N", [4, 16])
def test_diffuse_cubemap_matches_oracle(N):
    assert torch.cuda.is_available()
    from render_utils import diffuse_cubemap
    from oracle import cubemap_oracle as O
    g = torch.Generator().manual_seed(N)
    x = torch.rand(6, N, N, 3, generator=g).cuda().requires_grad_(True)
    G = torch.randn(6, N, N, 3, generator=g).cuda()
    out = diffuse_cubemap(x)
    (out * G).sum().backward()
    W = O.diffuse_matrix(N)
    want = W @ x.detach().cpu().double().numpy().reshape(-1, 3)
    gwant = W.T @ G.cpu().double().numpy().reshape(-1, 3)
    assert np.abs(out.detach().cpu().numpy().reshape(-1, 3) - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    assert np.abs(x.grad.cpu().numpy().reshape(-1, 3) - gwant).max() < 2e-5 * max(1.0, np.abs(gwant).max())


@pytest.mark.parametrize("N,roughness,cutoff", [(16, 0.5, 0.99), (16, 1.0, 0.99), (32, 0.27, 0.99), (32, 0.08, 0.9), (8, 0.04, 0.99)])
def test_specular_cubemap_matches_oracle(N, roughness, cutoff):
    assert torch.cuda.is_available()
    import render_utils as RU
    from oracle import cubemap_oracle as O
    g = torch.Generator().manual_seed(N + int(100 * roughness))
    x = torch.rand(6, N, N, 3, generator=g).cuda().requires_grad_(True)
    G = torch.randn(6, N, N, 3, generator=g).cuda()
    cos_cut = RU.ndf_cutoff(roughness, cutoff)
    raw = RU._specular_cubemap.apply(x, roughness, cos_cut)
    # texels whose direction sits within fp32 rounding of the cone boundary may fall on either side: compare with
    # the restatement evaluated with the boundary moved both ways
    Wlo, Whi = O.specular_matrix(N, roughness, cos_cut + 2e-6), O.specular_matrix(N, roughness, cos_cut - 2e-6)
    xs = x.detach().cpu().double().numpy().reshape(-1, 3)
    got = raw.detach().cpu().double().numpy().reshape(-1, 4)
    lo = np.concatenate([Wlo @ xs, Wlo.sum(1, keepdims=True)], axis=1)
    hi = np.concatenate([Whi @ xs, Whi.sum(1, keepdims=True)], axis=1)
    # the weight sum: fp32 sums of up to ~6000 terms; for lobes narrower than a texel the self weight 1 / (pi alpha^4)
    # comes from 1 - c^2 (1 - alpha^4) at c ~ 1, which fp32 resolves to ~1e-2 relative at alpha^4 = 4e-5 -- in the reference as well
    assert (got[:, 3] >= lo[:, 3] * (1 - 2e-2) - 1e-4).all() and (got[:, 3] <= hi[:, 3] * (1 + 2e-2) + 1e-4).all()
    # what the operator returns is the ratio, where that factor cancels
    ratio, rlo, rhi = got[:, :3] / got[:, 3:], lo[:, :3] / lo[:, 3:], hi[:, :3] / hi[:, 3:]
    assert (ratio >= np.minimum(rlo, rhi) - 2e-4).all() and (ratio <= np.maximum(rlo, rhi) + 2e-4).all()
    if np.abs(hi - lo).max() < 1e-12:      # no texel on the boundary: the gradient must match the transpose exactly
        out = RU.specular_cubemap(x, roughness, cutoff)
        (out * G).sum().backward()
        col, ws = lo[:, :3], lo[:, 3:]
        Gn = G.cpu().double().numpy().reshape(-1, 3)
        gwant = Wlo.T @ (Gn / ws)          # d(col / ws)/dx: ws does not depend on x
        assert np.abs(x.grad.cpu().numpy().reshape(-1, 3) - gwant).max() < 2e-4 * max(1.0, np.abs(gwant).max())
        assert np.abs(out.detach().cpu().numpy().reshape(-1, 3) - col / ws).max() < 1e-4


@pytest.mark.parametrize("N,roughness", [(256, 0.155), (128, 0.27), (64, 0.385), (1024, 0.12)])
def test_specular_tile_kernels_match_oracle_on_sampled_texels(N, roughness):
    """The levels of a 512^2 light that run the one-wave-per-8x8-tile kernels (csrc/cubemap.hip: specular_tile_kernel), at
    their own resolution and roughness: forward and backward against the restatement's weights for sampled texels (face
    centres, edges, corners and random ones) -- the full matrix has (6 N^2)^2 entries.  (1024, 0.12): a union box wider than one
    64-texel segment."""
    assert torch.cuda.is_available()
    import render_utils as RU
    from oracle import cubemap_oracle as O
    g = torch.Generator().manual_seed(N)
    x = torch.rand(6, N, N, 3, generator=g).cuda().requires_grad_(True)
    G = torch.randn(6, N, N, 4, generator=g).cuda()
    cos_cut = RU.ndf_cutoff(roughness, 0.99)
    raw = RU._specular_cubemap.apply(x, roughness, cos_cut)
    (gx,) = torch.autograd.grad(raw, x, G)
    T = 6 * N * N
    special = [0, N - 1, N * N - 1, (N // 2) * N + N // 2, 3 * N * N + 5, 5 * N * N + (N - 1) * N, 2 * N * N + 7 * N + N - 1, T - 1]
    rows = np.unique(np.concatenate([special, torch.randint(0, T, (40 if N < 512 else 8,), generator=g).numpy()]))
    chunk = 8 if N < 512 else 2   # (chunk, 6 N^2, 3) doubles for the half vectors
    xs = x.detach().cpu().double().numpy().reshape(-1, 3)
    Gn = G.cpu().double().numpy().reshape(-1, 4)[:, :3]
    got = raw.detach().cpu().double().numpy().reshape(-1, 4)[rows]
    ggot = gx.cpu().double().numpy().reshape(-1, 3)[rows]
    areas = O.texel_areas(N)
    checked = 0
    for k in range(0, len(rows), chunk):
        r = rows[k:k + chunk]
        Wlo, Whi = O.specular_matrix(N, roughness, cos_cut + 2e-6, rows=r), O.specular_matrix(N, roughness, cos_cut - 2e-6, rows=r)
        lo = np.concatenate([Wlo @ xs, Wlo.sum(1, keepdims=True)], axis=1)
        hi = np.concatenate([Whi @ xs, Whi.sum(1, keepdims=True)], axis=1)
        gk = got[k:k + chunk]
        assert (gk[:, 3] >= lo[:, 3] * (1 - 1e-3) - 1e-5).all() and (gk[:, 3] <= hi[:, 3] * (1 + 1e-3) + 1e-5).all()
        ratio, rlo, rhi = gk[:, :3] / gk[:, 3:], lo[:, :3] / lo[:, 3:], hi[:, :3] / hi[:, 3:]
        assert (ratio >= np.minimum(rlo, rhi) - 2e-4).all() and (ratio <= np.maximum(rlo, rhi) + 2e-4).all()
        # backward: grad_x[t] = sum_o W[o, t] G[o] = area(t) sum_o K[t, o] G[o] with K symmetric: row t of W, re-weighted
        exact = np.abs(Whi - Wlo).max(axis=1) < 1e-300   # rows without a texel on the cone boundary
        WT = Wlo / areas[None, :] * areas[r][:, None]
        want = WT @ Gn
        for j in np.nonzero(exact)[0]:
            assert np.abs(ggot[k + j] - want[j]).max() < 2e-4 * max(1.0, np.abs(want[j]).max()), (N, r[j])
            assert np.abs(gk[j] - lo[j]).max() < 2e-4 * max(1.0, np.abs(lo[j]).max()), (N, r[j])
            checked += 1
    # (at 1024^2 a 3700-pair cone has ~200 texels on its rim and nearly every row has one inside the +-2e-6 band: there the
    # forward is checked by the bounds above and the backward by the adjoint identity of the next test)
    assert checked >= (len(rows) // 3 if N < 512 else 1), "most sampled texels have no pair exactly on the cone boundary"


def test_specular_cubemap_backward_is_the_adjoint_at_full_size():
    """512^2 as in the reference (roughness 0.04: a cone of a few texels) and 64^2 with a wide lobe: <g, S x> = <S^T g, x>,
    and a constant environment stays constant after the division by the weight sum."""
    assert torch.cuda.is_available()
    import render_utils as RU
    gen = torch.Generator().manual_seed(3)
    for N, r in ((512, 0.04), (64, 0.385), (128, 0.27), (256, 0.155), (1024, 0.12)):
        x = torch.rand(6, N, N, 3, generator=gen).cuda().requires_grad_(True)
        G = torch.randn(6, N, N, 4, generator=gen).cuda()
        raw = RU._specular_cubemap.apply(x, r, RU.ndf_cutoff(r, 0.99))
        lhs = (raw[..., :3].double() * G[..., :3].double()).sum().item()
        (gx,) = torch.autograd.grad(raw, x, G)
        rhs = (gx.double() * x.detach().double()).sum().item()
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (N, lhs, rhs)
        c = RU.specular_cubemap(torch.full((6, N, N, 3), 0.6, device="cuda"), r)
        assert (c - 0.6).abs().max().item() < 1e-5
