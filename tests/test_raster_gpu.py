"""Parity tests proper: the HIP path, called through the drop-in op surface (and therefore the
C ABI), against the CPU oracle on the same seeded inputs.  Tolerances are north_star's:
1e-4 abs on rendered buffers (relative to channel magnitude for the distance channel, whose
values reach 10), 1e-3 rel on gradients, integer artefacts bit-exact."""
import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu


def _require_gpu():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    import gs2m_native
    gs2m_native.lib()  # raises loudly if the extension is missing


def _check(oracle, sc, grads=True, tol=Hh.ABS_TOL_BUFFERS, **kw):
    f, gr = Hh.run_oracle(oracle, sc, backward=grads, **kw)
    out, g = Hh.run_hip(sc, backward=grads, **kw)
    assert np.array_equal(out["radii"], f.radii), "radii"
    Hh.assert_observe_close(out["observe"], f)
    Hh.assert_image_close("color", out["color"], f.color, tol=tol, oracle_fwd=f)
    for ch in range(10):
        scale = max(1.0, float(np.abs(f.buffer[ch]).max()))
        Hh.assert_image_close(f"buffer[{ch}]", out["buffer"][ch], f.buffer[ch], tol=tol, scale=scale, oracle_fwd=f)
    assert np.all(out["buffer"][sc["fc"]:] == 0), "channels >= feature_count must stay zero"
    if grads:
        for k, v in g.items():
            Hh.assert_grad_close(k, v, gr[k])
        if not kw:  # SH + scale/rotation path: the two halves of the backward, each element-wise
            Hh.assert_two_stage(oracle, f, gr, Hh.run_hip_sums(sc))
    return f, out


@pytest.mark.parametrize("fc", [0, 1, 5, 9, 10])
def test_feature_counts(oracle_lib, fc):
    _require_gpu()
    sc = Hh.make_scene(4000, 200, 120, seed=fc, fc=fc, scale_hi=0.05, bg=(0.3, 0.1, 0.2))
    _check(oracle_lib, sc)


@pytest.mark.parametrize("fc", [2, 3, 4, 6, 7, 8])
def test_feature_counts_padded_variants(oracle_lib, fc):
    _require_gpu()
    sc = Hh.make_scene(1500, 96, 80, seed=10 + fc, fc=fc, scale_hi=0.06)
    _check(oracle_lib, sc)


@pytest.mark.parametrize("W,H", [(16, 16), (17, 33), (250, 130), (1, 1), (640, 360)])
def test_image_sizes(oracle_lib, W, H):
    _require_gpu()
    sc = Hh.make_scene(3000, W, H, seed=W, fc=9, scale_hi=0.08, bg=(0.0, 0.5, 1.0))
    _check(oracle_lib, sc)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_degrees(oracle_lib, deg):
    _require_gpu()
    sc = Hh.make_scene(2000, 128, 96, seed=deg, fc=5, sh_degree=deg, scale_hi=0.05)
    _check(oracle_lib, sc)


def test_large_and_thin_gaussians(oracle_lib):
    """big splats (whole-tile coverage, long lists, early termination) and needle-like ones.
    For splats hundreds of pixels long the three terms of
    power = -0.5(A dx^2 + C dy^2) - B dx dy are O(1e2..1e3) and cancel to O(1), so ANY fp32
    evaluation order (the oracle's unfused one, the HIP kernel's FMA form, nvcc's contraction of the
    reference) carries ~1e-4 absolute noise in power; the HIP kernels therefore evaluate power in the
    written order (common.h gs2m_power) so that this case stays inside 1e-4 too.  DESIGN.md, Numerics."""
    _require_gpu()
    sc = Hh.make_scene(1500, 160, 128, seed=5, fc=10, scale_lo=0.0005, scale_hi=0.6, bg=(0.2, 0.2, 0.2))
    _check(oracle_lib, sc)


def test_many_instances_per_gaussian(oracle_lib):
    """screen-filling splats on a 20 x 16 tile image: hundreds of tile instances per Gaussian, i.e. Gaussians whose
    partial rows span several 64-instance windows of the row reduction and waves with more windows than it prefetches
    valid flags for (gaussian_bwd.hip: row_reduce_kernel), and tile lists many batches long."""
    _require_gpu()
    sc = Hh.make_scene(700, 320, 256, seed=11, fc=9, scale_lo=0.2, scale_hi=1.5, bg=(0.1, 0.0, 0.3))
    f, out = _check(oracle_lib, sc)
    assert f.tiles_touched.max() > 128 and f.num_rendered > 20000


@pytest.mark.parametrize("fc", [1, 5, 10])
def test_heavy_units_at_every_row_width(oracle_lib, fc):
    """the heavy units' row sums (heavy_reduce_kernel<3, 4, 6>: rows of 12, 16 and 24 floats; 20 floats: the test below)"""
    _require_gpu()
    sc = Hh.make_scene(3000, 512, 320, seed=50 + fc, fc=fc, scale_lo=0.003, scale_hi=0.03, bg=(0.1, 0.1, 0.0))
    big = torch.rand(3000, generator=torch.Generator().manual_seed(6)) < 0.03
    sc["g"]["scales"] = torch.where(big[:, None], sc["g"]["scales"] * 40.0, sc["g"]["scales"])
    f, out = _check(oracle_lib, sc)
    assert (f.tiles_touched >= 64).sum() >= 20, "the scene is meant to hold heavy Gaussians"


@pytest.mark.parametrize("refbin", [False, True])
def test_big_splats_take_the_heavy_unit_paths(oracle_lib, refbin):
    """Gaussians with hundreds of tile instances (here: screen-filling splats on a 48 x 27 tile image, several per emit wave, next to
    thousands of small ones): HEAVY (common.h: GS2M_HEAVY_TILES) -- expanded a unit of 64 instances per wave by emit_heavy_kernel, four
    gradient rows reserved per instance, added up per unit by heavy_reduce_kernel, the units' sums fetched by the whole wave in
    gaussian_bwd_kernel -- every check of the ordinary scenes, in both binning modes, plus bitwise reproducibility of the gradients
    (fixed summation orders in these paths too)."""
    _require_gpu()
    import gs2m_native
    sc = Hh.make_scene(6000, 768, 432, seed=31, fc=9, scale_lo=0.003, scale_hi=0.03, bg=(0.05, 0.1, 0.2))
    big = torch.rand(6000, generator=torch.Generator().manual_seed(4)) < 0.02
    sc["g"]["scales"] = torch.where(big[:, None], sc["g"]["scales"] * 60.0, sc["g"]["scales"])
    gs2m_native.set_reference_binning(refbin)
    f, out = _check(oracle_lib, sc)
    assert (f.tiles_touched >= 512).sum() >= 20, "the scene is meant to hold big Gaussians"
    a = Hh.run_hip_sums(sc)
    b = Hh.run_hip_sums(sc)
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_crowded_waves_hand_medium_splats_to_the_heavy_units(oracle_lib):
    """A trained model keeps its medium-sized splats together in index order (a densification generation): a wave of 64 consecutive
    Gaussians holding more than 320 instances between them is CROWDED and hands everything from 8 tiles on to the heavy units
    (common.h: gs2m_heavy).  Here: the first 1500 of 8000 Gaussians cover 8 to 39 tiles each."""
    _require_gpu()
    import gs2m_native
    import diff_gaussian_rasterization as dgr
    sc = Hh.make_scene(8000, 640, 400, seed=41, fc=9, scale_lo=0.003, scale_hi=0.02, bg=(0.0, 0.1, 0.0))
    sc["g"]["scales"][:1500] *= 10.0
    sc["g"]["scales"][2000::97] *= 8.0  # lone medium splats in ordinary waves
    f, out = _check(oracle_lib, sc)
    tt = np.asarray(f.tiles_touched).astype(np.int64)
    g = {k: v.cuda() for k, v in sc["g"].items()}
    st = Hh.settings_for(sc, "cuda")
    e = torch.Tensor([])
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, sc["H"], sc["W"], g["shs"], sc["sh_degree"], st.campos, False, sc["fc"])
    torch.cuda.synchronize()
    lay = gs2m_native.debug_layout(8000, R, sc["W"], sc["H"])
    al = (-geomB.data_ptr()) % 256
    gr = geomB[al + lay.gauss_rows: al + lay.gauss_rows + 4 * 8000].cpu().numpy().view(np.uint32)
    ttd = geomB[al + lay.tiles_touched: al + lay.tiles_touched + 4 * 8000].cpu().numpy().view(np.uint32).astype(np.int64)
    heavy = ((gr & np.uint32(0x80000000)) != 0) & (ttd > 0)
    assert int((heavy & (ttd < 40)).sum()) > 200, "the scene is meant to hold crowded waves"
    assert int((~heavy & (ttd >= 8))[1600:].sum()) > 0, "... next to ordinary ones that keep their medium splats"
    a = Hh.run_hip_sums(sc)
    b = Hh.run_hip_sums(sc)
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_uniformly_medium_scene_switches_the_crowded_rule_off(oracle_lib):
    """Every Gaussian covers a dozen tiles and more: every wave is crowded, there is no imbalance to repair, and a unit of 256 rows per
    Gaussian would be 10x the rows the frame needs.  The forward notices (heavy units x 256 > 6 x num_rendered), counts the units again
    without the crowded-wave rule and tells the emit kernel (api.hip): only Gaussians of 40 tiles and more stay heavy."""
    _require_gpu()
    import gs2m_native
    import diff_gaussian_rasterization as dgr
    sc = Hh.make_scene(3000, 640, 400, seed=43, fc=9, scale_lo=0.04, scale_hi=0.09, bg=(0.0, 0.0, 0.1))
    f, out = _check(oracle_lib, sc)
    g = {k: v.cuda() for k, v in sc["g"].items()}
    st = Hh.settings_for(sc, "cuda")
    e = torch.Tensor([])
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, sc["H"], sc["W"], g["shs"], sc["sh_degree"], st.campos, False, sc["fc"])
    torch.cuda.synchronize()
    lay = gs2m_native.debug_layout(3000, R, sc["W"], sc["H"])
    al = (-geomB.data_ptr()) % 256
    gr = geomB[al + lay.gauss_rows: al + lay.gauss_rows + 4 * 3000].cpu().numpy().view(np.uint32)
    ttd = geomB[al + lay.tiles_touched: al + lay.tiles_touched + 4 * 3000].cpu().numpy().view(np.uint32).astype(np.int64)
    U = int(geomB[al + lay.counters: al + lay.counters + 16].cpu().numpy().view(np.uint32)[3])
    heavy = ((gr & np.uint32(0x80000000)) != 0) & (ttd > 0)
    w = np.concatenate([ttd, np.zeros((-3000) % 64, np.int64)]).reshape(-1, 64)
    assert ((w * (w < 40)).sum(1) > 320).mean() > 0.5, "the scene is meant to have most waves crowded"
    assert np.array_equal(heavy, ttd >= 40) and U == int(((ttd[heavy] + 63) // 64).sum()), "the crowded-wave rule is off for this frame"
    assert 256 * U <= 6 * R + 65536


def test_dense_scene_terminates(oracle_lib):
    """many opaque layers: exercises T < 1e-4 termination, n_contrib < list length, block early-out."""
    _require_gpu()
    sc = Hh.make_scene(20000, 96, 64, seed=6, fc=9, scale_lo=0.02, scale_hi=0.2)
    sc["g"]["opacities"] = torch.clamp(sc["g"]["opacities"] * 2.0, max=0.999)
    f, out = _check(oracle_lib, sc)
    assert (f.final_T < 1e-3).mean() > 0.3


def test_precomputed_colors_and_cov(oracle_lib):
    _require_gpu()
    sc = Hh.make_scene(2500, 128, 128, seed=7, fc=9, scale_hi=0.05)
    g = torch.Generator().manual_seed(1)
    colors = torch.rand(2500, 3, generator=g)
    import gs2m_scene
    prm = gs2m_scene.GaussianParams.from_activated(
        sc["g"]["means3D"], sc["g"]["shs"], sc["g"]["scales"], sc["g"]["rotations"], sc["g"]["opacities"],
        torch.full((2500, 3), 0.5), torch.full((2500, 1), 0.5), torch.full((2500, 1), 0.5))
    cov = prm.get_covariance().contiguous()
    _check(oracle_lib, sc, colors_precomp=colors, cov3D_precomp=cov)


def test_non_contiguous_camera_matrices(oracle_lib):
    """The reference Camera builds world_view_transform as torch.tensor(...).transpose(0, 1).cuda()
    (scene/cameras.py:64): a NON-contiguous (4,4) tensor, and full_proj_transform from it.  Forward and backward must
    both see the logical matrix (the reference binding calls .contiguous() in both directions); a rotated, translated
    camera makes the transposed read differ."""
    _require_gpu()
    import gs2m_synth as S
    cam = S.look_at_camera(160, 120, eye=(1.5, -0.7, 0.5), target=(0.0, 0.0, 6.0))
    sc = Hh.make_scene(2500, 160, 120, seed=31, fc=9, scale_hi=0.06, cam=cam)
    f, gr = Hh.run_oracle(oracle_lib, sc)
    for k in ("viewmatrix", "projmatrix"):
        m = cam[k]
        cam[k] = m.t().contiguous().t()  # same values, strides (1, 4)
        assert not cam[k].is_contiguous() and torch.equal(cam[k], m)
    out, g = Hh.run_hip(sc)
    Hh.assert_image_close("color", out["color"], f.color, oracle_fwd=f)
    for k, v in g.items():
        Hh.assert_grad_close(k, v, gr[k])


def test_empty_and_invisible(oracle_lib):
    _require_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = Hh.make_scene(64, 64, 48, seed=8, fc=9, bg=(0.25, 0.5, 0.75))
    # P = 0: background only
    st = Hh.settings_for(sc, "cuda")
    z = lambda *s: torch.zeros(*s, device="cuda")
    color, radii, observe, buffer = GaussianRasterizer(st)(z(0, 3), z(0, 4), z(0, 1), shs=z(0, 16, 3), scales=z(0, 3),
                                                           rotations=z(0, 4), features=z(0, 10))
    assert radii.numel() == 0 and observe.numel() == 0
    assert torch.allclose(color, sc["bg"].cuda()[:, None, None].expand_as(color))
    assert torch.all(buffer == 0)
    # everything behind the camera: same image, zero gradients, radii 0
    sc["g"]["means3D"][:, 2] = -sc["g"]["means3D"][:, 2].abs() - 1.0
    f, out = _check(oracle_lib, sc)
    assert f.num_rendered == 0 and np.all(out["radii"] == 0)


def test_argument_errors():
    _require_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = Hh.make_scene(8, 32, 32)
    r = GaussianRasterizer(Hh.settings_for(sc, "cuda"))
    g = {k: v.cuda() for k, v in sc["g"].items()}
    m2 = torch.zeros(8, 4, device="cuda")
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(g["means3D"], m2, g["opacities"], scales=g["scales"], rotations=g["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(g["means3D"], m2, g["opacities"], shs=g["shs"], scales=g["scales"])
    with pytest.raises(RuntimeError, match="HIP"):
        r(g["means3D"].cpu(), m2, g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
          features=g["features"])


def test_prefiltered_is_a_checked_promise():
    """`prefiltered=True` (settings tuple, __init__.py:153): identical outputs while every Gaussian is in front of the near
    plane; with one behind it the reference traps the device (auxiliary.h:155-158) -- here the forward raises and the
    context stays usable."""
    _require_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = Hh.make_scene(500, 96, 64, seed=12, fc=9, behind_frac=0.0)
    g = {k: v.cuda() for k, v in sc["g"].items()}
    m2 = torch.zeros(500, 4, device="cuda")
    st = Hh.settings_for(sc, "cuda")
    args = dict(shs=g["shs"], scales=g["scales"], rotations=g["rotations"], features=g["features"])
    ref = GaussianRasterizer(st)(g["means3D"], m2, g["opacities"], **args)
    pre = GaussianRasterizer(st._replace(prefiltered=True))(g["means3D"], m2, g["opacities"], **args)
    for a, b in zip(ref, pre):
        assert torch.equal(a, b)
    bad = g["means3D"].clone()
    bad[17, 2] = 0.1  # view z <= 0.2
    with pytest.raises(RuntimeError, match="should have been prefiltered"):
        GaussianRasterizer(st._replace(prefiltered=True))(bad, m2, g["opacities"], **args)
    again = GaussianRasterizer(st)(bad, m2, g["opacities"], **args)  # the same scene without the promise renders
    assert int(again[1][17]) == 0 and torch.isfinite(again[0]).all()


def test_binning_artefacts_bit_exact(oracle_lib):
    """radii, tiles_touched, depth keys, the sorted (tile, depth) list, ranges and n_contrib against the
    oracle, including ties in depth (stability of both sorts)."""
    _require_gpu()
    import gs2m_native
    import diff_gaussian_rasterization as dgr
    gs2m_native.set_reference_binning(True)
    try:
        _binning_bit_exact(oracle_lib, gs2m_native, dgr)
    finally:
        gs2m_native.set_reference_binning(False)


def _binning_bit_exact(oracle_lib, gs2m_native, dgr):
    sc = Hh.make_scene(6000, 320, 200, seed=9, fc=9, scale_hi=0.06)
    m = sc["g"]["means3D"]
    m[:, 2] = torch.round(m[:, 2] * 4) / 4  # many exactly equal depths -> ties resolved by Gaussian id
    m[:100, 2] = -1.0
    f, _ = Hh.run_oracle(oracle_lib, sc, backward=False)
    g = {k: v.cuda() for k, v in sc["g"].items()}
    st = Hh.settings_for(sc, "cuda")
    e = torch.Tensor([])
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, sc["H"], sc["W"], g["shs"], 3, st.campos, False, 9)
    torch.cuda.synchronize()
    P, W, H = 6000, sc["W"], sc["H"]
    assert R == f.num_rendered
    lay = gs2m_native.debug_layout(P, R, W, H)
    al = lambda t: (-t.data_ptr()) % 256
    view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
    vis = f.radii > 0
    assert np.array_equal(radii.cpu().numpy(), f.radii)
    assert np.array_equal(view(geomB, lay.tiles_touched, P, np.uint32), f.tiles_touched)
    dk = view(geomB, lay.depth_key, P, np.uint32)
    assert np.array_equal(dk[vis], f.depths[vis].view(np.uint32)) and np.all(dk[~vis] == 0xFFFFFFFF)
    rec = view(geomB, lay.rec, P * 32, np.float32).reshape(P, 32)
    assert np.array_equal(rec[vis, 0:2], f.means2D[vis])
    pl = view(binB, lay.point_list, R, np.uint32) & np.uint32(0x0FFFFFFF)  # list-driven kernels: quadrant mask above the id
    tk = view(binB, lay.tile_keys, R, np.uint32)
    assert np.array_equal(pl, f.vals_sorted), "sorted Gaussian ids"
    keys = (tk.astype(np.uint64) << np.uint64(32)) | f.depths[pl].view(np.uint32).astype(np.uint64)
    assert np.array_equal(keys, f.keys_sorted), "sorted (tile<<32 | depth) keys"
    assert np.all(np.diff(keys.astype(np.int64)) >= 0)
    Tn = f.tiles_x * f.tiles_y
    assert np.array_equal(view(imgB, lay.ranges, 2 * Tn, np.uint32).reshape(Tn, 2), f.ranges)
    nc = view(imgB, lay.n_contrib, W * H, np.uint32).reshape(H, W)
    assert (nc != f.n_contrib).mean() <= 1e-4
    Hh.assert_n_contrib_close(nc, f)  # every pixel whose last contributor differs sits on a threshold of the blend
    assert np.all(nc <= (f.ranges[:, 1] - f.ranges[:, 0]).reshape(f.tiles_y, f.tiles_x).repeat(16, 0).repeat(16, 1)[:H, :W])


def test_default_binning_is_a_safe_subset(oracle_lib):
    """default mode drops tiles a Gaussian cannot reach with alpha >= 1/255: each tile list must be a
    subsequence of the reference's list (same order) that still holds every instance with at least one
    contributing pixel (so all outputs are unchanged -- checked by every other test in this file)."""
    _require_gpu()
    import gs2m_native
    import diff_gaussian_rasterization as dgr
    P, W, H = 3000, 160, 112
    sc = Hh.make_scene(P, W, H, seed=15, fc=9, scale_lo=0.003, scale_hi=0.08)
    f, _ = Hh.run_oracle(oracle_lib, sc, backward=False)
    g = {k: v.cuda() for k, v in sc["g"].items()}
    st = Hh.settings_for(sc, "cuda")
    e = torch.Tensor([])
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, H, W, g["shs"], 3, st.campos, False, 9)
    torch.cuda.synchronize()
    assert 0 < R <= f.num_rendered
    assert np.array_equal(radii.cpu().numpy(), f.radii), "radii keep the reference value"
    lay = gs2m_native.debug_layout(P, R, W, H)
    al = lambda t: (-t.data_ptr()) % 256
    view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
    pl = view(binB, lay.point_list, R, np.uint32) & np.uint32(0x0FFFFFFF)  # list-driven kernels: quadrant mask above the id
    Tn = f.tiles_x * f.tiles_y
    rg = view(imgB, lay.ranges, 2 * Tn, np.uint32).reshape(Tn, 2)
    dropped = 0
    for t in range(Tn):
        ref = f.vals_sorted[f.ranges[t, 0]:f.ranges[t, 1]]
        mine = pl[rg[t, 0]:rg[t, 1]]
        it = iter(ref.tolist())
        assert all(any(x == y for y in it) for x in mine.tolist()), f"tile {t}: not a subsequence of the reference list"
        gone = np.setdiff1d(ref, mine)
        dropped += len(gone)
        if len(gone):  # the dropped instances must not contribute anywhere in this tile
            tx, ty = t % f.tiles_x, t // f.tiles_x
            px, py = np.meshgrid(np.arange(tx * 16, min(tx * 16 + 16, W)), np.arange(ty * 16, min(ty * 16 + 16, H)))
            xy = f.means2D[gone].astype(np.float64); co = f.conic_opacity[gone].astype(np.float64)
            dx = xy[:, 0, None, None] - px[None]; dy = xy[:, 1, None, None] - py[None]
            power = -0.5 * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
            alpha = np.minimum(0.99, co[:, 3, None, None] * np.exp(np.minimum(power, 0)))
            assert not ((power <= 0) & (alpha >= 1.0 / 255.0 * (1 - 1e-4))).any(), f"tile {t}: dropped a contributing instance"
    assert dropped > 0, "the scene should exercise the tile-rectangle shrink"


def test_mark_visible(oracle_lib):
    _require_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = Hh.make_scene(5000, 64, 64, seed=11, behind_frac=0.3)
    vis = GaussianRasterizer(Hh.settings_for(sc, "cuda")).markVisible(sc["g"]["means3D"].cuda()).cpu().numpy()
    ref = oracle_lib.mark_visible(sc["g"]["means3D"].numpy(), sc["cam"]["viewmatrix"].numpy(), sc["cam"]["projmatrix"].numpy())
    assert vis.dtype == np.bool_ and np.array_equal(vis, ref)


def test_backward_is_bitwise_reproducible():
    _require_gpu()
    sc = Hh.make_scene(5000, 256, 144, seed=12, fc=9, scale_hi=0.05)
    _, g1 = Hh.run_hip(sc)
    _, g2 = Hh.run_hip(sc)
    for k in g1:
        assert np.array_equal(g1[k], g2[k]), k


def test_gradcheck_against_golden(oracle_lib):
    """committed golden vectors (oracle outputs, tests/golden/make_golden.py) vs the HIP path."""
    _require_gpu()
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "raster_small.npz")
    z = np.load(path)
    sc = Hh.scene_from_golden(z)
    out, g = Hh.run_hip(sc)
    assert np.array_equal(out["radii"], z["radii"])
    Hh.assert_image_close("color", out["color"], z["color"])
    Hh.assert_image_close("buffer", out["buffer"], z["buffer"], scale=10.0)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "features"):
        Hh.assert_grad_close(k, g[k], z["grad_" + k])


def test_render_counterpart_matches_oracle(oracle_lib):
    """gaussian_renderer.render() on the device vs the same pre/post-processing around the oracle."""
    _require_gpu()
    import gs2m_scene
    from gaussian_renderer import render
    sc = Hh.make_scene(3000, 160, 120, seed=13, fc=9, scale_hi=0.06)
    out_dev, out_ref = Hh.render_pair(oracle_lib, sc, material_stage=True, blend_metallic=True, sobel_normal=True)
    for k in ("render", "alpha_map", "distance_map", "normal_map", "albedo_map", "roughness_map", "metallic_map",
              "local_normal_map"):
        scale = max(1.0, float(np.abs(out_ref[k]).max()))
        Hh.assert_image_close(k, out_dev[k], out_ref[k], scale=scale)
    assert np.array_equal(out_dev["radii"], out_ref["radii"])
    assert out_dev["viewspace_points"].shape == (3000, 4)
    assert set(out_dev.keys()) >= {"render", "viewspace_points", "visibility_filter", "radii", "observe", "alpha_map",
                                   "distance_map", "depth_map", "normal_map", "albedo_map", "roughness_map",
                                   "metallic_map", "normal_mask", "local_normal_map", "sobel_map"}


def test_render_end_to_end_gradients(oracle_lib):
    """autograd through render()'s pre/post-processing AND the rasterizer: raw-parameter gradients on the
    device vs the same Python code around the oracle-backed op on the CPU."""
    _require_gpu()
    sc = Hh.make_scene(2000, 128, 96, seed=14, fc=9, scale_hi=0.06)
    dev, ref = Hh.render_pair(oracle_lib, sc, grads=True, material_stage=True, blend_metallic=True)
    names = ["xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic"]
    for n, a, b in zip(names, dev["param_grads"], ref["param_grads"]):
        assert Hh.rel_err(a, b) < 2e-3, (n, Hh.rel_err(a, b))
    assert Hh.rel_err(dev["viewspace_grad"], ref["viewspace_grad"]) < 1e-3


def test_debug_mode_and_markers_do_not_change_results():
    """gs2m_set_debug(1): stream synchronize + error check after every stage (a fault would be raised naming the
    stage); gs2m_set_markers(1): roctx ranges around the stages.  Same bits out, and scratch release works."""
    _require_gpu()
    import gs2m_native
    import diff_gaussian_rasterization as dgr
    sc = Hh.make_scene(3000, 160, 96, seed=41, fc=9, scale_hi=0.06)
    out0, g0 = Hh.run_hip(sc)
    gs2m_native.set_debug(True)
    gs2m_native.set_markers(True)
    out1, g1 = Hh.run_hip(sc)
    gs2m_native.set_debug(False)
    gs2m_native.set_markers(False)
    for k in ("color", "buffer", "radii", "observe"):
        assert np.array_equal(out0[k], out1[k]), k
    for k in g0:
        assert np.array_equal(g0[k], g1[k]), k
    assert gs2m_native.lib().gs2m_stage_name(8) == b"blend_bwd"
    dgr.release_scratch()
    out2, g2 = Hh.run_hip(sc)  # the scratch is simply allocated again
    assert np.array_equal(g0["means3D"], g2["means3D"])


def test_two_forwards_before_their_backwards_and_a_retained_graph():
    """The autograd path reuses one binning buffer per device from call to call, but only when nothing references it any
    more: a second view rendered before the first one's backward (multi_view_loss does that) and a backward repeated
    with retain_graph must see their own forward's state."""
    _require_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    import gs2m_synth as S
    sc_a = Hh.make_scene(3000, 160, 96, seed=51, fc=9, scale_hi=0.06)
    sc_b = Hh.make_scene(3000, 160, 96, seed=51, fc=9, scale_hi=0.06, cam=S.look_at_camera(160, 96, eye=(0.8, 0.2, 0.0), target=(0.0, 0.0, 6.0)))
    _, ga = Hh.run_hip(sc_a)
    _, gb = Hh.run_hip(sc_b)

    def fwd(sc):
        g = {k: v.cuda().requires_grad_(True) for k, v in sc["g"].items()}
        m2 = torch.zeros(3000, 4, device="cuda", requires_grad=True)
        color, _, _, buffer = GaussianRasterizer(Hh.settings_for(sc, "cuda"))(g["means3D"], m2, g["opacities"], shs=g["shs"], scales=g["scales"],
                                                                          rotations=g["rotations"], features=g["features"])
        return g, (color * sc["Gc"].cuda()).sum() + (buffer * sc["Gb"].cuda()).sum()
    pa, la = fwd(sc_a)
    pb, lb = fwd(sc_b)          # second forward while the first one's graph is alive
    la.backward(retain_graph=True)
    pc, lc = fwd(sc_b)          # third forward while the first graph is retained
    lb.backward()
    first = pa["means3D"].grad.clone()
    pa["means3D"].grad = None
    la.backward()               # the retained graph again: same gradients
    lc.backward()
    assert np.array_equal(first.cpu().numpy(), ga["means3D"]) and torch.equal(pa["means3D"].grad, first)
    assert np.array_equal(pb["means3D"].grad.cpu().numpy(), gb["means3D"])
    assert np.array_equal(pc["means3D"].grad.cpu().numpy(), gb["means3D"])


@pytest.mark.parametrize("split", [False, True])
def test_backward_without_colour_gradient_skips_dL_dSH(split):
    """A view whose colour output receives no gradient (the multi-view term's neighbour view: only its depth and normal maps enter
    the loss): dL/dcolour is identically zero, so the op hands autograd None for the SH tensor(s) and the per-Gaussian kernel does
    not write them (dL_dshs = NULL, include/gs2m_raster.h) -- every other gradient bit for bit what an explicit all-zero colour
    gradient gives."""
    _require_gpu()
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = Hh.make_scene(20_000, 320, 200, seed=4, fc=5)
    st = Hh.settings_for(sc, "cuda")
    Gb = sc["Gb"].cuda()

    def run(with_zero_colour_term):
        g = {k: v.cuda().requires_grad_(True) for k, v in sc["g"].items()}
        m2 = torch.zeros(g["means3D"].shape[0], 4, device="cuda", requires_grad=True)
        kw = {}
        if split:
            dc = g["shs"][:, :1].detach().clone().contiguous().requires_grad_(True)
            rest = g["shs"][:, 1:].detach().clone().contiguous().requires_grad_(True)
            kw = dict(shs=dc, shs_rest=rest)
            sh_leaves = (dc, rest)
        else:
            kw = dict(shs=g["shs"])
            sh_leaves = (g["shs"],)
        color, radii, observe, buffer = GaussianRasterizer(st)(g["means3D"], m2, g["opacities"], scales=g["scales"], rotations=g["rotations"],
                                                               features=g["features"], **kw)
        loss = (buffer * Gb).sum()
        if with_zero_colour_term:
            loss = loss + (color * 0.0).sum()
        loss.backward()
        return [g[k].grad for k in ("means3D", "opacities", "scales", "rotations", "features")] + [m2.grad], [t.grad for t in sh_leaves]

    a, sha = run(False)
    b, shb = run(True)
    assert all(t is None for t in sha), "no gradient for the SH tensors when the colour output got none"
    assert all(t is not None and float(t.abs().max()) == 0.0 for t in shb), "an all-zero colour gradient gives all-zero dL/dSH"
    for x, y in zip(a, b):
        assert x is not None and torch.equal(x, y)
