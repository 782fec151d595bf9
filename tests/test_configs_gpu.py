"""Every BASELINE.json configuration at its stated size (tensor sizes table), the HIP path against the
CPU oracle on the WHOLE workload (the oracle does one 1M / 1080p view, forward + backward, in about five seconds on the
GPU box's host cores):
  C1  10k Gaussians, 256x256, feature_count 10        (the oracle itself is checked against CPU autograd at this size in
                                                      tests/test_oracle.py)
  C2  500k Gaussians, 1920x1080, feature_count 5
  C3  1M Gaussians, 1080p, feature_count 9            the bench workload; also its "deferred pbr.shade" leg: the fused
                                                      shading on the 1080p G-buffer of that render vs the op-by-op form
  C5  2M Gaussians, 1080p, feature_count 9            the per-GPU shape of the 8-GPU configuration, on one GPU (the 8-GPU leg
                                                      itself needs the driver's node; tests/test_dp.py covers the sharding
                                                      logic on two gloo ranks)
each in BOTH binning modes: radii exact, observe and images with the threshold-event proof, gradients element-wise, the
backward in its two halves (helpers.assert_two_stage); in reference-binning mode also the sorted instance list, the tile
ranges and n_contrib against the oracle's, bit for bit.  Plus size-independent properties (compositing identities,
reproducibility).  C4 (train.py loop on a COLMAP-format scene) is tests/test_train_gpu.py."""
import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu


def _forward(sc, bg):
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = dict(sc); sc["bg"] = torch.tensor(bg, dtype=torch.float32)
    g = {k: v.cuda() for k, v in sc["g"].items()}
    m2 = torch.zeros(g["means3D"].shape[0], 4, device="cuda")
    with torch.no_grad():
        return GaussianRasterizer(Hh.settings_for(sc, "cuda"))(g["means3D"], m2, g["opacities"], shs=g["shs"],
                                                              scales=g["scales"], rotations=g["rotations"],
                                                              features=g["features"])


def _properties(sc, P, fc):
    """compositing identities, radii / observe consistency, bitwise reproducible gradients"""
    c0, radii, obs, b0 = _forward(sc, (0.0, 0.0, 0.0))
    c1, _, _, b1 = _forward(sc, (1.0, 1.0, 1.0))
    T = c1 - c0  # = final transmittance, per channel
    assert torch.allclose(T[0], T[1], atol=2e-6) and torch.allclose(T[0], T[2], atol=2e-6)
    assert float(T.min()) >= -1e-6 and float(T.max()) <= 1 + 1e-6
    assert torch.allclose(b0[0] + T[0], torch.ones_like(T[0]), atol=2e-5), "sum of blend weights = 1 - T"
    assert torch.equal(b0, b1), "the G-buffer has no background term"
    assert torch.all(b0[fc:] == 0)
    assert int((radii > 0).sum()) > 0.7 * P
    assert bool(torch.all((obs > 0) <= (radii > 0))) and int(obs.sum()) > 0
    del c0, c1, b0, b1, T
    _, g1 = Hh.run_hip(sc)
    _, g2 = Hh.run_hip(sc)
    for k in g1:
        assert np.all(np.isfinite(g1[k])), k
        assert np.array_equal(g1[k], g2[k]), f"{k} not bitwise reproducible"
    assert float(np.abs(g1["means3D"]).max()) > 0


_ORACLE_CACHE = {}


def _oracle_for(oracle, key, sc):
    """one oracle forward + backward per configuration, shared by the two binning modes"""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE.clear()  # one configuration at a time: an OracleForward of the 2M scene holds ~1 GB
        _ORACLE_CACHE[key] = Hh.run_oracle(oracle, sc)
    return _ORACLE_CACHE[key]


def _against_oracle(oracle, key, sc, reference_binning):
    import gs2m_native
    import diff_gaussian_rasterization as dgr
    f, gr = _oracle_for(oracle, key, sc)
    gs2m_native.set_reference_binning(reference_binning)
    out, g = Hh.run_hip(sc)
    assert np.array_equal(out["radii"], f.radii)
    Hh.assert_observe_close(out["observe"], f)
    Hh.assert_image_close("color", out["color"], f.color, oracle_fwd=f)
    for ch in range(10):
        scale = max(1.0, float(np.abs(f.buffer[ch]).max()))
        Hh.assert_image_close(f"buffer[{ch}]", out["buffer"][ch], f.buffer[ch], scale=scale, oracle_fwd=f)
    assert np.all(out["buffer"][sc["fc"]:] == 0)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "features"):
        Hh.assert_grad_close(k, g[k], gr[k])
    sums = Hh.run_hip_sums(sc)
    Hh.assert_chain_exceptions_conditioned(f, g, gr, sums)  # the end-to-end exceptions of dL/dscale, dL/drot are ill-conditioned Gaussians
    del out, g
    Hh.assert_two_stage(oracle, f, gr, sums)
    if reference_binning:  # the integer artefacts of the benched workload itself, bit for bit
        P, W, H = f.P, sc["W"], sc["H"]
        gd = {k: v.cuda() for k, v in sc["g"].items()}
        st = Hh.settings_for(sc, "cuda")
        e = torch.Tensor([])
        R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
            st.bg, gd["means3D"], e, gd["opacities"], gd["scales"], gd["rotations"], 1.0, e, gd["features"], st.viewmatrix,
            st.projmatrix, st.tanfovx, st.tanfovy, H, W, gd["shs"], 3, st.campos, False, sc["fc"])
        torch.cuda.synchronize()
        assert R == f.num_rendered
        lay = gs2m_native.debug_layout(P, R, W, H)
        al = lambda t: (-t.data_ptr()) % 256
        view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
        assert np.array_equal(view(geomB, lay.tiles_touched, P, np.uint32), f.tiles_touched)
        assert np.array_equal(view(binB, lay.point_list, R, np.uint32) & np.uint32(0x0FFFFFFF), f.vals_sorted), "sorted Gaussian ids"
        assert np.array_equal(view(binB, lay.tile_keys, R, np.uint32), (f.keys_sorted >> np.uint64(32)).astype(np.uint32)), "sorted tile ids"
        Tn = f.tiles_x * f.tiles_y
        assert np.array_equal(view(imgB, lay.ranges, 2 * Tn, np.uint32).reshape(Tn, 2), f.ranges)
        nc = view(imgB, lay.n_contrib, W * H, np.uint32).reshape(H, W)
        assert (nc != f.n_contrib).mean() <= 1e-4
        Hh.assert_n_contrib_close(nc, f)  # every pixel whose last contributor differs sits on a threshold of the blend
    gs2m_native.set_reference_binning(False)
    dgr.release_scratch()


def test_config_c1(oracle_lib):
    """10k random Gaussians, 1 camera, 256x256, feature_count 10: the same scene tests/test_oracle.py checks the oracle
    on against CPU autograd"""
    sc = Hh.make_scene(10_000, 256, 256, seed=1, fc=10)
    for refbin in (False, True):
        _against_oracle(oracle_lib, "c1", sc, refbin)


@pytest.mark.parametrize("reference_binning", [False, True], ids=["default_binning", "reference_binning"])
def test_config_c2_full_size_against_oracle(oracle_lib, reference_binning):
    """500k synthetic Gaussians, 1920x1080, colour + depth + normal buffers (feature_count 5): the whole view vs the oracle"""
    sc = Hh.make_scene(500_000, 1920, 1080, seed=0, fc=5)
    if not reference_binning:
        _properties(sc, 500_000, 5)
    _against_oracle(oracle_lib, "c2", sc, reference_binning)


@pytest.mark.parametrize("reference_binning", [False, True], ids=["default_binning", "reference_binning"])
def test_config_c3_full_size_against_oracle(oracle_lib, reference_binning):
    """1M Gaussians, 1920x1080, feature_count 9 -- the bench workload itself (bench.py --config c3) vs the oracle"""
    sc = Hh.make_scene(1_000_000, 1920, 1080, seed=0, fc=9)
    _against_oracle(oracle_lib, "c3", sc, reference_binning)


@pytest.mark.parametrize("reference_binning", [False, True], ids=["default_binning", "reference_binning"])
def test_config_c5_per_gpu_shape_full_size_against_oracle(oracle_lib, reference_binning):
    """2M Gaussians, 1920x1080, feature_count 9 on ONE GPU (every rank of the 8-GPU configuration holds this): the whole
    view vs the oracle"""
    P = 2_000_000
    sc = Hh.make_scene(P, 1920, 1080, seed=0, fc=9)
    if not reference_binning:
        _properties(sc, P, 9)
    _against_oracle(oracle_lib, "c5", sc, reference_binning)


def test_config_c3_deferred_shading_leg():
    """1M Gaussians, --material on: the albedo / roughness / normal G-buffers of the 1080p render go through the deferred
    shading (pbr/shade.py); the fused kernel pair against the op-by-op form on that very G-buffer, outputs and the
    gradients back to the G-buffer and the light."""
    import gs2m_synth as S
    from gaussian_renderer import render
    from gs2m_scene import Camera, GaussianParams, PipelineParams
    from pbr import CubemapLight, get_brdf_lut, pbr_shading, pbr_shading_fused
    P, W, H = 1_000_000, 1920, 1080
    dev = "cuda"
    cam0 = S.make_camera(W, H)
    g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam0, seed=0).items()}
    u = lambda c, s: torch.rand(P, c, generator=torch.Generator().manual_seed(s)).to(dev) * 0.8 + 0.1
    pc = GaussianParams.from_activated(g["means3D"], g["shs"], g["scales"], g["rotations"], g["opacities"].clamp(0.01, 0.99),
                                       u(3, 1), u(1, 2), u(1, 3))
    cam = Camera(cam0, dev)
    with torch.no_grad():
        out = render(cam, pc, PipelineParams(), torch.zeros(3, device=dev), material_stage=True)
    n = out["normal_map"].permute(1, 2, 0).contiguous()
    rough = out["roughness_map"].permute(1, 2, 0).clamp(0.04, 1.0).contiguous()
    v = -torch.nn.functional.normalize(cam.get_rays(), dim=-1).reshape(H, W, 3).contiguous()
    lut = get_brdf_lut().to(dev)
    Gw = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    res = {}
    for fused in (False, True):
        torch.manual_seed(11)
        light = CubemapLight(base_res=128, device=dev)
        with torch.no_grad():
            light.base.copy_(torch.rand(light.base.shape, generator=torch.Generator().manual_seed(12)).to(dev) * 1.5)
        albedo = out["albedo_map"].permute(1, 2, 0).contiguous().clone().requires_grad_(True)
        light.build_mips()
        if fused:
            pkg = pbr_shading_fused(light, n, v, albedo, rough, metallic=None, brdf_lut=lut)
        else:
            pkg = pbr_shading(light, n, v, albedo, rough, metallic=None, occlusion=torch.ones_like(rough),
                              irradiance=torch.zeros_like(rough), brdf_lut=lut)
        (pkg["render_rgb"] * Gw).sum().backward()
        res[fused] = (pkg["render_rgb"].detach().reshape(H, W, 3), albedo.grad, light.base.grad)
    assert (res[True][0] - res[False][0]).abs().max().item() < 5e-5
    assert (res[True][1] - res[False][1]).abs().max().item() < 1e-4 * max(1.0, res[False][1].abs().max().item())
    gb0, gb1 = res[False][2], res[True][2]
    assert (gb1 - gb0).abs().max().item() < 5e-4 * max(1.0, gb0.abs().max().item())
    assert res[False][0].abs().max().item() > 0.05
