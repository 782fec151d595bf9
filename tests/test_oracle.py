"""CPU tests of the oracle (test infrastructure): against the reference's importable helpers
(golden vectors), against a PyTorch-autograd restatement, against its own committed outputs,
and structural invariants of the binning artefacts."""
import os

import numpy as np
import pytest
import torch

import gs2m_synth as S
import helpers as Hh

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_sh_forward_matches_reference_eval_sh(oracle_lib):
    """oracle SH->RGB (CR/forward.cu:20-67) vs vectors produced by the reference's utils/sh_utils.eval_sh."""
    z = np.load(os.path.join(GOLD, "ref_helpers.npz"))
    sh = z["sh_coeffs"]  # (P,3,16) reference layout -> (P,16,3) rasterizer layout
    dirs = z["sh_dirs"]
    P = sh.shape[0]
    cam = S.make_camera(64, 64)
    # place the Gaussians along the golden directions from the camera so that normalize(mean - campos) == dirs
    depth = 4.0 / np.maximum(dirs[:, 2:3], 0.2)
    means = (dirs * np.abs(depth)).astype(np.float32)
    means[:, 2] = np.abs(means[:, 2]) + 1.0
    d = means / np.linalg.norm(means, axis=1, keepdims=True)
    sys_path_ref = torch.tensor(sh)
    for deg in range(4):
        f = oracle_lib.forward(means, np.full((P, 1), 0.5, np.float32), shs=np.ascontiguousarray(sh.transpose(0, 2, 1)),
                               scales=np.full((P, 3), 0.05, np.float32), rotations=np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32),
                               bg=np.zeros(3, np.float32), viewmatrix=cam["viewmatrix"].numpy(),
                               projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(), W=64, H=64,
                               tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=deg)
        import gs2m_scene
        ref = gs2m_scene.eval_sh(deg, sys_path_ref, torch.tensor(d)).numpy()
        vis = f.radii > 0
        assert vis.sum() > 10
        want = np.maximum(ref + 0.5, 0.0)
        assert np.allclose(f.rgb[vis], want[vis], atol=2e-6), deg
        assert np.array_equal(f.clamped[vis].astype(bool), (ref[vis] + 0.5) < 0)
        # and the python restatement used by render() equals the reference's eval_sh on the golden dirs
        mine = gs2m_scene.eval_sh(deg, sys_path_ref, torch.tensor(dirs)).numpy()
        assert np.allclose(mine, z[f"sh_eval_deg{deg}"], atol=1e-6)


def test_camera_matrices_match_reference_helpers():
    z = np.load(os.path.join(GOLD, "ref_helpers.npz"))
    for i in range(z["cam_R"].shape[0]):
        v = S.world2view(z["cam_R"][i], z["cam_T"][i])
        assert np.array_equal(v, z["cam_view"][i])
        p = S.projection_matrix(0.01, 100.0, float(z["cam_fov"][i, 0]), float(z["cam_fov"][i, 1])).numpy()
        assert np.array_equal(p, z["cam_proj"][i])


def test_sobel_normals_match_reference_helper():
    import gs2m_scene
    z = np.load(os.path.join(GOLD, "ref_helpers.npz"))
    d, K, E = torch.tensor(z["nd_depth"]), torch.tensor(z["nd_K"]), torch.tensor(z["nd_E"])
    assert np.allclose(gs2m_scene.normal_from_depth_image(d, K, E, view_space=False).numpy(), z["nd_world"], atol=1e-6)
    assert np.allclose(gs2m_scene.normal_from_depth_image(d, K, E, view_space=True).numpy(), z["nd_view"], atol=1e-6)


def test_oracle_matches_committed_golden(oracle_lib):
    z = np.load(os.path.join(GOLD, "raster_small.npz"))
    sc = Hh.scene_from_golden(z)
    f, gr = Hh.run_oracle(oracle_lib, sc)
    assert f.num_rendered == int(z["num_rendered"])
    assert np.array_equal(f.radii, z["radii"]) and np.array_equal(f.vals_sorted, z["vals_sorted"])
    assert np.array_equal(f.ranges, z["ranges"]) and np.array_equal(f.n_contrib, z["n_contrib"])
    assert np.array_equal(f.observe, z["observe"])
    assert np.allclose(f.color, z["color"], atol=1e-6) and np.allclose(f.buffer, z["buffer"], atol=1e-5)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "features", "cov3D", "conics", "colors"):
        assert Hh.rel_err(gr[k], z["grad_" + k]) < 1e-5, k


@pytest.mark.parametrize("seed,bg", [(1, (0.2, 0.3, 0.4)), (2, (0.0, 0.0, 0.0))])
def test_oracle_vs_autograd(oracle_lib, seed, bg):
    """hand-written backward incl. quirks Q1-Q5 vs float64 autograd with the quirks patched in."""
    from torch_ref import rasterize_dense
    W, H, P, fc = 64, 48, 300, 10
    sc = Hh.make_scene(P, W, H, seed=seed, fc=fc, scale_lo=0.01, scale_hi=0.15, bg=bg)
    f, gr = Hh.run_oracle(oracle_lib, sc)
    cam = sc["cam"]
    d = {k: v.double().requires_grad_(True) for k, v in sc["g"].items()}
    color, buf, radii, aux = rasterize_dense(
        d["means3D"], d["opacities"], d["shs"], None, d["scales"], d["rotations"], None, d["features"], bg=sc["bg"],
        viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"], W=W, H=H,
        tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3, feature_count=fc)
    assert np.array_equal(radii.numpy(), f.radii)
    assert np.abs(color.detach().numpy() - f.color).max() < 1e-4
    assert np.abs(buf.detach().numpy() - f.buffer).max() < 1e-3
    ((color * sc["Gc"].double()).sum() + (buf * sc["Gb"].double()).sum()).backward()
    for k in ("means3D", "opacities", "shs", "scales", "rotations", "features"):
        assert Hh.rel_err(gr[k], d[k].grad.numpy()) < 1e-3, k


def test_oracle_vs_autograd_precomputed(oracle_lib):
    from torch_ref import rasterize_dense
    import gs2m_scene
    W, H, P, fc = 48, 48, 200, 5
    sc = Hh.make_scene(P, W, H, seed=4, fc=fc, scale_lo=0.02, scale_hi=0.2)
    prm = gs2m_scene.GaussianParams.from_activated(
        sc["g"]["means3D"], sc["g"]["shs"], sc["g"]["scales"], sc["g"]["rotations"], sc["g"]["opacities"],
        torch.full((P, 3), 0.5), torch.full((P, 1), 0.5), torch.full((P, 1), 0.5))
    cov = prm.get_covariance().contiguous()
    cols = torch.rand(P, 3, generator=torch.Generator().manual_seed(3))
    f, gr = Hh.run_oracle(oracle_lib, sc, colors_precomp=cols, cov3D_precomp=cov)
    cam = sc["cam"]
    m = sc["g"]["means3D"].double().requires_grad_(True)
    o = sc["g"]["opacities"].double().requires_grad_(True)
    c = cols.double().requires_grad_(True)
    cv = cov.double().requires_grad_(True)
    ft = sc["g"]["features"].double().requires_grad_(True)
    color, buf, radii, _ = rasterize_dense(m, o, None, c, None, None, cv, ft, bg=sc["bg"], viewmatrix=cam["viewmatrix"],
                                           projmatrix=cam["projmatrix"], campos=cam["campos"], W=W, H=H,
                                           tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=0, feature_count=fc)
    ((color * sc["Gc"].double()).sum() + (buf * sc["Gb"].double()).sum()).backward()
    assert Hh.rel_err(gr["colors"], c.grad.numpy()) < 1e-3
    assert Hh.rel_err(gr["cov3D"], cv.grad.numpy()) < 1e-3
    assert Hh.rel_err(gr["means3D"], m.grad.numpy()) < 1e-3


def test_binning_invariants(oracle_lib):
    """sorted keys monotone on the sorted bits, ranges partition [0,R), n_contrib <= range
    length, radii > 0 <=> tiles_touched > 0, ties keep Gaussian-id order."""
    sc = Hh.make_scene(5000, 200, 136, seed=5, fc=1, scale_hi=0.08)
    sc["g"]["means3D"][:, 2] = torch.round(sc["g"]["means3D"][:, 2] * 2) / 2
    f, _ = Hh.run_oracle(oracle_lib, sc, backward=False)
    assert f.sort_bits == 32 + oracle_lib.higher_msb(f.tiles_x * f.tiles_y)
    k = f.keys_sorted
    assert np.all(k[1:] >= k[:-1])
    same = k[1:] == k[:-1]
    assert same.any() and np.all(f.vals_sorted[1:][same] > f.vals_sorted[:-1][same])
    r = f.ranges
    touched = r[:, 1] > r[:, 0]
    starts = r[touched, 0]; ends = r[touched, 1]
    order = np.argsort(starts)
    assert starts[order][0] == 0 and ends[order][-1] == f.num_rendered and np.all(starts[order][1:] == ends[order][:-1])
    assert np.all(r[~touched] == 0)
    assert np.array_equal(f.radii > 0, f.tiles_touched > 0)
    assert f.point_offsets[-1] == f.num_rendered
    lens = (r[:, 1] - r[:, 0]).reshape(f.tiles_y, f.tiles_x).repeat(16, 0).repeat(16, 1)[:f.H, :f.W]
    assert np.all(f.n_contrib <= lens)


def test_knn_oracle_small():
    from oracle import oracle
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [5, 5, 5]], np.float32)
    d = oracle.knn_dist2(pts)
    assert np.isclose(d[0], (1 + 4 + 9) / 3.0)
    brute = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    np.fill_diagonal(brute, np.inf)
    assert np.allclose(d, np.sort(brute, 1)[:, :3].mean(1))


def test_torch_ref_gradcheck():
    """finite differences (fp64) on the autograd restatement itself, tiny scene."""
    from torch_ref import rasterize_dense
    sc = Hh.make_scene(6, 16, 16, seed=3, fc=2, scale_lo=0.3, scale_hi=0.6, behind_frac=0.0)
    cam = sc["cam"]
    g = sc["g"]

    def fn(means, opac, scales):
        color, buf, _, _ = rasterize_dense(means, opac, g["shs"].double(), None, scales, g["rotations"].double(), None,
                                           g["features"].double(), bg=sc["bg"], viewmatrix=cam["viewmatrix"],
                                           projmatrix=cam["projmatrix"], campos=cam["campos"], W=16, H=16,
                                           tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3, feature_count=2)
        return (color * sc["Gc"].double()).sum() + (buf * sc["Gb"].double()).sum()

    m = g["means3D"].double().requires_grad_(True)
    o = (g["opacities"].double() * 0.5).requires_grad_(True)  # keep alpha below the 0.99 clamp (Q1)
    s = g["scales"].double().requires_grad_(True)
    loss = fn(m, o, s)
    go, = torch.autograd.grad(loss, o)
    eps = 1e-6
    for i in range(3):
        op = o.detach().clone(); om = o.detach().clone()
        op[i, 0] += eps; om[i, 0] -= eps
        fd = (fn(m, op, s) - fn(m, om, s)) / (2 * eps)
        assert abs(fd.item() - go[i, 0].item()) <= 1e-4 * max(1.0, abs(fd.item()))


def test_pixel_walk_used_to_explain_outliers_reproduces_the_oracle(oracle_lib):
    """helpers.pixel_threshold_events (the per-pixel walk the GPU parity tests use to prove that an out-of-tolerance
    pixel sits on an alpha = 1/255 or T = 1e-4 threshold) follows the oracle's blend: same final T and the same
    last contributor on every sampled pixel that is not itself at a threshold."""
    import helpers as Hh
    sc = Hh.make_scene(3000, 96, 80, seed=21, fc=9, scale_hi=0.08)
    f, _ = Hh.run_oracle(oracle_lib, sc, backward=False)
    rng = np.random.default_rng(0)
    n_ok = 0
    for _ in range(300):
        x, y = int(rng.integers(0, 96)), int(rng.integers(0, 80))
        ev, T, last = Hh.pixel_threshold_events(f, x, y, full=True)
        if ev <= 1e-5:
            continue  # the walk evaluates exp in double: a pixel AT a threshold may legitimately differ
        assert abs(T - float(f.final_T[y, x])) <= 2e-6 * max(1.0, T), (x, y, T, f.final_T[y, x])
        assert last == int(f.n_contrib[y, x]), (x, y)
        n_ok += 1
    assert n_ok > 250


def test_config_c1_oracle_vs_cpu_autograd(oracle_lib):
    """BASELINE.json configs[0] at its stated size: 10k random Gaussians, 1 camera, 256x256, feature_count 10 -- the
    oracle's forward and hand-written backward against the PyTorch CPU autograd rasterizer (float64, composited tile
    by tile), "plumbing + grad check" as the config says.  The same scene is run on the GPU against the oracle in
    tests/test_configs_gpu.py::test_config_c1."""
    from torch_ref import rasterize_dense
    P, W, H, fc = 10_000, 256, 256, 10
    sc = Hh.make_scene(P, W, H, seed=1, fc=fc)
    f, gr = Hh.run_oracle(oracle_lib, sc)
    cam = sc["cam"]
    d = {k: v.double().requires_grad_(True) for k, v in sc["g"].items()}
    color, buf, radii, aux = rasterize_dense(
        d["means3D"], d["opacities"], d["shs"], None, d["scales"], d["rotations"], None, d["features"], bg=sc["bg"],
        viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"], W=W, H=H,
        tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3, feature_count=fc, tiled=True)
    assert np.array_equal(radii.numpy(), f.radii)
    # 1e-4 on the images; a pixel outside it must sit on an alpha = 1/255 / T = 1e-4 threshold (fp64 vs fp32 power)
    Hh.assert_image_close("color", color.detach().numpy(), f.color, oracle_fwd=f)
    for ch in range(fc):
        scale = max(1.0, float(np.abs(f.buffer[ch]).max()))
        Hh.assert_image_close(f"buffer[{ch}]", buf[ch].detach().numpy(), f.buffer[ch], scale=scale, oracle_fwd=f)
    ((color * sc["Gc"].double()).sum() + (buf * sc["Gb"].double()).sum()).backward()
    # the oracle evaluates the reference's fp32 formulas; against exact (float64) derivatives its elements carry the
    # fp32 conditioning of the projection / covariance chains, and a pixel whose alpha sits on 1/255 contributes in one
    # arithmetic and not in the other (one such pixel moves means3D by 2e-3 of the tensor's maximum here), so:
    # max-norm 5e-3 on every tensor, element-wise 1e-3 (+ 1e-4 of the tensor's rms) on all but a counted 1 %
    for k in ("means3D", "opacities", "shs", "scales", "rotations", "features"):
        ref = d[k].grad.numpy()
        frac, worst, _ = Hh.grad_stats(gr[k], ref, 1e-3, 1e-4)
        assert worst < 5e-3, (k, worst)
        assert frac <= 1e-2, (k, frac)
