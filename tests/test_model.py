"""Host-side model logic on the CPU: the schedule / conversions against vectors generated from the
reference's own helpers (tests/golden/make_golden.py), the PLY layout, the optimizer surgery bookkeeping.  The
densification / PLY / COLMAP parity tests run on the CPU (`-m "not gpu"`) AND on the device (`-m gpu`: the model's tensors,
the fused Adam's state and the split's torch.normal draws then live on the GPU, as in training)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_model.npz")
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def test_lr_schedule_and_conversions_match_reference_vectors():
    from gs2m_model import expon_lr, rgb_to_sh
    from gs2m_scene import inverse_sigmoid
    d = np.load(GOLD)
    for (a, b, c, dm, e), want in zip(d["lr_cfgs"], d["lr_values"]):
        f = expon_lr(a, b, lr_delay_steps=int(c), lr_delay_mult=dm, max_steps=int(e))
        got = np.array([f(int(s)) for s in d["lr_steps"]])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    x = torch.tensor(d["unit_x"])
    assert np.array_equal(inverse_sigmoid(x).numpy(), d["inverse_sigmoid"])
    assert np.array_equal(rgb_to_sh(x).numpy(), d["rgb2sh"])


def _random_model(n, seed=0):
    from gs2m_model import GaussianModel
    g = torch.Generator().manual_seed(seed)
    m = GaussianModel(3, device="cpu")
    r = lambda *s: torch.randn(*s, generator=g)
    m.parameterize((r(n, 3), r(n, 1, 3), r(n, 15, 3), r(n, 3), r(n, 4), r(n, 1), r(n, 3), r(n, 1), r(n, 1)))
    return m


def test_ply_layout_and_round_trip(tmp_path):
    from gs2m_model import GaussianModel
    m = _random_model(37)
    path = str(tmp_path / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    header, body = raw.split(b"end_header\n", 1)
    lines = header.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    assert names == m.construct_list_of_attributes() and len(names) == 6 + 3 + 45 + 1 + 3 + 4 + 3 + 2   # GM:260-278
    assert len(body) == 37 * len(names) * 4
    table = np.frombuffer(body, dtype="<f4").reshape(37, len(names))
    assert np.array_equal(table[:, 0:3], m._xyz.detach().numpy()) and not table[:, 3:6].any()
    # SH tensors are stored channel-major: f_rest_{c*15+k} = features_rest[:, k, c]   (GM:283-284)
    assert np.array_equal(table[:, 9 + 1 * 15 + 4], m._features_rest.detach().numpy()[:, 4, 1])
    assert np.array_equal(table[:, 6 + 2], m._features_dc.detach().numpy()[:, 0, 2])
    again = GaussianModel(3, device="cpu")
    again.load_ply(path)
    for a, b in zip(m.parameters(), again.parameters()):
        assert a.shape == b.shape and torch.equal(a.detach(), b.detach())
    assert again.active_sh_degree == 3
    # ascii files and extra elements after the vertex element are read too
    v = GaussianModel.read_ply_vertices(path)
    apath = str(tmp_path / "ascii.ply")
    with open(apath, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex 37\n" + "".join(f"property float {n}\n" for n in names)
                + "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for i in range(37):
            f.write(" ".join(repr(float(v[n][i])) for n in names) + "\n")
    third = GaussianModel(3, device="cpu")
    third.load_ply(apath)
    for a, b in zip(m.parameters(), third.parameters()):
        assert torch.equal(a.detach(), b.detach())


def test_prune_and_cat_keep_parameters_and_adam_state_aligned():
    """_prune_optimizer / cat_tensors_to_optimizer (GM:388-455) with torch.optim.Adam standing in for the fused
    optimizer (same state layout; the fused step itself needs the GPU)."""
    from gs2m_model import OptimizationParams
    m = _random_model(50)
    m.spatial_lr_scale = 1.0
    class Args(OptimizationParams):
        prune_init_points = False          # (on by default as in the reference: it would drop the largest points here)

    m.training_setup(Args, optimizer_cls=torch.optim.Adam)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    m.optimizer.step()
    before = {g["name"]: (g["params"][0].detach().clone(), m.optimizer.state[g["params"][0]]["exp_avg"].clone()) for g in m.optimizer.param_groups}
    mask = torch.arange(50) % 5 == 0
    m.prune_points(mask)
    assert m.get_xyz.shape[0] == 40 and m.denom.shape == (40, 1) and m.max_radii2D.shape == (40,)
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        st = m.optimizer.state[p]
        assert torch.equal(p.detach(), before[g["name"]][0][~mask]) and torch.equal(st["exp_avg"], before[g["name"]][1][~mask])
        assert float(st["step"]) == 1.0
    m.densification_postfix(**{g["name"]: torch.full((7,) + tuple(g["params"][0].shape[1:]), 2.0) for g in m.optimizer.param_groups})
    assert m.get_xyz.shape[0] == 47 and m._opacity is [g for g in m.optimizer.param_groups if g["name"] == "opacity"][0]["params"][0]
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        st = m.optimizer.state[p]
        assert torch.equal(p.detach()[40:], torch.full_like(p.detach()[40:], 2.0)) and not st["exp_avg"][40:].any() and not st["exp_avg_sq"][40:].any()
        assert torch.equal(st["exp_avg"][:40], before[g["name"]][1][~mask])
    m.reset_opacity()
    assert (m.get_opacity <= 0.01 + 1e-6).all() and not m.optimizer.state[m._opacity]["exp_avg"].any()


@pytest.mark.parametrize("device", DEVICES)
def test_colmap_reader_matches_the_reference_reader(tmp_path, device):
    """tests/golden/colmap_small/*.bin parsed by gs2m_colmap against what the reference's own reader returned for the
    same files (tests/golden/colmap_small.npz, generated by make_golden.py), plus a write -> read round trip."""
    import gs2m_colmap as C
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = np.load(os.path.join(here, "colmap_small.npz"))
    folder = os.path.join(here, "colmap_small")
    cams = C.read_intrinsics_binary(os.path.join(folder, "cameras.bin"))
    assert sorted(cams) == list(d["cam_ids"])
    for cid, c in cams.items():
        assert c.model == str(d[f"cam{cid}_model"]) and [c.width, c.height] == list(d[f"cam{cid}_wh"])
        assert np.array_equal(c.params, d[f"cam{cid}_params"])
    imgs = C.read_extrinsics_binary(os.path.join(folder, "images.bin"))
    assert list(imgs) == list(d["img_ids"])
    for iid, im in imgs.items():
        assert np.array_equal(im.qvec, d[f"img{iid}_qvec"]) and np.array_equal(im.tvec, d[f"img{iid}_tvec"])
        assert im.camera_id == int(d[f"img{iid}_cam"]) and im.name == str(d[f"img{iid}_name"])
        assert np.array_equal(im.xys.reshape(-1, 2), d[f"img{iid}_xys"]) and np.array_equal(im.point3D_ids, d[f"img{iid}_p3d"])
        assert np.array_equal(np.transpose(C.qvec2rotmat(im.qvec)), d[f"img{iid}_R"])
    infos = C.colmap_cameras(imgs, cams)
    assert [i.Fx for i in infos][:2] == [700.5, 910.0] and [i.Fy for i in infos][:2] == [701.25, 910.0]   # PINHOLE fx, fy; SIMPLE_PINHOLE f, f
    norm = C.nerf_normalization(infos)
    np.testing.assert_allclose(norm["translate"], d["norm_translate"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(norm["radius"], float(d["norm_radius"]), rtol=1e-12)
    xyz, rgb, err = C.read_points3D_binary(os.path.join(folder, "points3D.bin"))
    assert np.array_equal(xyz, d["pts_xyz"]) and np.array_equal(rgb, d["pts_rgb"]) and np.array_equal(err, d["pts_err"])
    # round trip through the writer, and quaternion <-> matrix
    C.write_model(str(tmp_path), cams.values(), imgs.values(), xyz, rgb, err)
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        assert open(os.path.join(folder, name), "rb").read() == open(os.path.join(str(tmp_path), name), "rb").read()
    for im in imgs.values():
        np.testing.assert_allclose(C.rotmat2qvec(C.qvec2rotmat(im.qvec)), im.qvec, atol=1e-12)
    # the cameras the training loop builds from these records (gs2m_train.load_colmap_dataset), on `device`: W2C rotation and
    # translation are the file's (scene/dataset_readers.py:81-82 stores R transposed), the centre is -R^T t
    import gs2m_synth as S
    from gs2m_scene import Camera
    for info, iid in zip(infos, imgs):
        cam = Camera(S.make_camera(info.width, info.height, fx=info.Fx, fy=info.Fy, R=info.R, T=info.T), device)
        assert cam.world_view_transform.device.type == device and cam.full_proj_transform.device.type == device
        w2c = cam.world_view_transform.t().cpu().numpy()
        Rw2c = np.transpose(d[f"img{iid}_R"])
        np.testing.assert_allclose(w2c[:3, :3], Rw2c, atol=1e-6)
        np.testing.assert_allclose(w2c[:3, 3], d[f"img{iid}_tvec"], atol=1e-5)
        np.testing.assert_allclose(cam.camera_center.cpu().numpy(), -Rw2c.T @ d[f"img{iid}_tvec"], atol=1e-4)


def test_hyperparameter_defaults_match_the_reference():
    """gs2m_model.OptimizationParams, gs2m_mvs.MultiViewParams and gs2m_scene.PipelineParams against the defaults of the
    reference's own argument classes (tests/golden/ref_defaults.json, generated by make_golden.py)."""
    import json
    import gs2m_model, gs2m_mvs, gs2m_scene
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_defaults.json")))
    seen = 0
    for cls, ref in ((gs2m_model.OptimizationParams, gold["OptimizationParams"]), (gs2m_mvs.MultiViewParams, gold["OptimizationParams"]),
                     (gs2m_scene.PipelineParams, gold["PipelineParams"])):
        for k, v in vars(cls).items():
            if k.startswith("_") or k in ("split_sh", "fused_render_ops", "fused_activations", "fused_loss_tail"):   # this repository's additions
                continue
            assert k in ref, f"{cls.__name__}.{k} is not a reference parameter"
            assert ref[k] == v, f"{cls.__name__}.{k} = {v}, reference {ref[k]}"
            seen += 1
    assert seen >= 50


def test_blender_dataset_reader(tmp_path):
    """transforms_train.json (camera-to-world, OpenGL axes) -> the reference's camera convention: a camera written from known COLMAP-style
    (R, T) must come back with the same view matrix and centre; RGBA images are composited on the background."""
    import json
    from PIL import Image as PILImage
    import gs2m_synth as S
    import gs2m_train
    W, H, fx = 64, 48, 70.0
    frames, want = [], []
    os.makedirs(tmp_path / "train")
    for k, eye in enumerate(((0.0, 0.0, -4.0), (3.0, 1.0, -2.0), (-2.0, -1.5, 3.0))):
        cam = S.look_at_camera(W, H, eye, (0.0, 0.0, 0.0), fx=fx)
        w2c = np.eye(4)
        w2c[:3, :3], w2c[:3, 3] = cam["R"].T, cam["T"]
        c2w = np.linalg.inv(w2c)
        c2w[:3, 1:3] *= -1                      # COLMAP axes -> OpenGL axes (the reader flips them back)
        frames.append({"file_path": f"train/r_{k}", "transform_matrix": c2w.tolist()})
        want.append(cam)
        rgba = np.zeros((H, W, 4), dtype=np.uint8)
        rgba[:, : W // 2] = (255, 0, 0, 255)    # left half opaque red, right half transparent
        PILImage.fromarray(rgba, "RGBA").save(tmp_path / "train" / f"r_{k}.png")
    json.dump({"camera_angle_x": 2 * np.arctan(W / (2 * fx)), "frames": frames}, open(tmp_path / "transforms_train.json", "w"))
    for white in (False, True):
        cams, gts, xyz, cols, radius = gs2m_train.load_blender_dataset(str(tmp_path), white_background=white, n_points=500, device="cpu")
        for c, w in zip(cams, want):
            assert (c.image_width, c.image_height) == (W, H) and abs(c.Fx - fx) < 1e-4
            assert (c.world_view_transform - w["viewmatrix"]).abs().max().item() < 1e-5
            assert (c.camera_center - w["campos"]).abs().max().item() < 1e-5
        assert gts[0].shape == (3, H, W) and gts[0][0, 0, 0].item() == 1.0
        assert gts[0][:, 0, W - 1].tolist() == ([1.0, 1.0, 1.0] if white else [0.0, 0.0, 0.0])
        assert xyz.shape == (500, 3) and np.abs(xyz).max() <= 1.3 and radius > 0


# ---------------------------------------------------------------------------------------------------------------------
# densification / pruning / opacity reset / Adam-state surgery against the REFERENCE'S OWN CLASS (scene/gaussian_model.py:362-573):
# tests/golden/ref_densify.npz holds every tensor of the reference's GaussianModel before and after each operation
# (tests/golden/make_densify_golden.py imports the class from /root/reference in the build container and drives it on the CPU)
DENSIFY_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_densify.npz")
_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "albedo", "roughness", "metallic")


@pytest.mark.parametrize("device", DEVICES)
def test_densify_prune_reset_match_the_reference_class(device, monkeypatch):
    """clone + split + prune with the Adam-moment surgery, an opacity reset, a second round with max_screen_size, reduce_opacity,
    prune_points, add_densification_stats and prune_init_points: gs2m_model.GaussianModel against the reference class's own
    tensors after every step -- bit for bit on the CPU; on the GPU (the product's fused Adam, whose state the surgery edits in
    training) the same Gaussians selected at every step and values to 1e-6 (the device's exp / log / sigmoid).  The split's
    torch.normal draws are the reference run's, replayed: the product must ask for them with the reference's `std`."""
    from gs2m_model import GaussianModel, OptimizationParams
    z = np.load(DENSIFY_GOLD)
    max_grad, max_grad_abs, min_opacity, extent, percent_dense, lr_scale = (float(v) for v in z["args"])
    t = lambda a: torch.tensor(a).to(device)
    model = GaussianModel(3, device)
    model.spatial_lr_scale = lr_scale
    model.parameterize([t(z[f"s0/p/{k}"]) for k in ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic")])

    class Opt(OptimizationParams):
        prune_init_points = False
    Opt.percent_dense = percent_dense
    model.training_setup(Opt, optimizer_cls=torch.optim.Adam if device == "cpu" else None)
    # the optimizer's moments: a step on zero gradients creates the state without moving anything, then the reference's values go in
    for grp in model.optimizer.param_groups:
        grp["params"][0].grad = torch.zeros_like(grp["params"][0])
    model.optimizer.step()
    model.optimizer.zero_grad(set_to_none=True)
    for grp in model.optimizer.param_groups:
        p = grp["params"][0]
        assert torch.equal(p.detach(), t(z[f"s0/p/{grp['name']}"])), "a step on zero gradients moves nothing"
        st = model.optimizer.state[p]
        st["exp_avg"].copy_(t(z[f"s0/m/{grp['name']}"]))
        st["exp_avg_sq"].copy_(t(z[f"s0/v/{grp['name']}"]))

    def same(a, want, what):
        want = t(want)
        assert a.shape == want.shape, (what, tuple(a.shape), tuple(want.shape))
        if device == "cpu":
            assert torch.equal(a, want), what
        else:
            torch.testing.assert_close(a, want, rtol=1e-6, atol=1e-7, msg=lambda m: f"{what}: {m}")

    def check(tag):
        assert model.get_xyz.shape[0] == z[f"{tag}/p/xyz"].shape[0], tag
        attrs = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling", rotation="_rotation",
                     albedo="_albedo", roughness="_roughness", metallic="_metallic")
        for grp in model.optimizer.param_groups:
            k, p = grp["name"], grp["params"][0]
            assert getattr(model, attrs[k]) is p, (tag, k, "the model's attribute is the optimizer's parameter")
            same(p.detach(), z[f"{tag}/p/{k}"], (tag, "param", k))
            st = model.optimizer.state.get(p)
            assert st is not None, (tag, k)
            same(st["exp_avg"], z[f"{tag}/m/{k}"], (tag, "exp_avg", k))
            same(st["exp_avg_sq"], z[f"{tag}/v/{k}"], (tag, "exp_avg_sq", k))
        same(model.xyz_gradient_accum, z[f"{tag}/accum"], (tag, "accum"))
        same(model.xyz_gradient_accum_abs, z[f"{tag}/accum_abs"], (tag, "accum_abs"))
        same(model.denom, z[f"{tag}/denom"], (tag, "denom"))
        same(model.max_radii2D, z[f"{tag}/max_radii"], (tag, "max_radii"))

    check("s0")
    draws = []
    def replay_normal(*a, **k):
        i = len(draws)
        assert not a and set(k) == {"mean", "std"} and float(k["mean"].abs().max()) == 0.0
        same(k["std"].detach(), z[f"normal{i}/std"], ("split", i, "std of the draw"))
        draws.append(i)
        return t(z[f"normal{i}/out"])
    monkeypatch.setattr(torch, "normal", replay_normal)
    stage = 0
    for rnd, screen in ((0, None), (1, 20)):
        model.xyz_gradient_accum, model.xyz_gradient_accum_abs = t(z[f"in{rnd}/accum"]), t(z[f"in{rnd}/accum_abs"])
        model.denom, model.max_radii2D = t(z[f"in{rnd}/denom"]), t(z[f"in{rnd}/max_radii"])
        model.densify_and_prune(max_grad, max_grad_abs, min_opacity, extent, screen)
        stage += 1
        check(f"s{stage}")
        if rnd == 0:
            model.reset_opacity()
            stage += 1
            check(f"s{stage}")
    assert draws == [0, 1], "one draw per split"
    model.reduce_opacity()
    check("s4")
    model.prune_points(t(z["in5/mask"]))
    check("s5")
    vs = torch.zeros(model.get_xyz.shape[0], 4, device=device, requires_grad=True)
    vs.grad = t(z["in6/grad"])
    model.add_densification_stats(vs, t(z["in6/filter"]))
    model.add_densification_stats(vs, t(z["in6/filter"]))
    check("s6")
    model.prune_init_points()
    check("s7")


@pytest.mark.parametrize("device", DEVICES)
def test_ply_bytes_match_the_reference_layout(tmp_path, device):
    """tests/golden/model_small.ply was assembled byte by byte in the layout GaussianModel.save_ply of the reference
    produces (tests/golden/make_ply_golden.py): our writer must produce the same bytes from the same tensors, and our
    reader must recover the tensors from it."""
    from gs2m_model import GaussianModel
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "model_small.npz"))
    model = GaussianModel(3, device)
    model.parameterize([torch.tensor(z[k]) for k in ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic")])
    out = tmp_path / "out.ply"
    model.save_ply(str(out))
    assert out.read_bytes() == open(os.path.join(here, "model_small.ply"), "rb").read()
    back = GaussianModel(3, device)
    back.load_ply(os.path.join(here, "model_small.ply"))
    assert back._xyz.device.type == device
    for k, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                    ("rotation", "_rotation"), ("albedo", "_albedo"), ("roughness", "_roughness"), ("metallic", "_metallic")):
        assert np.array_equal(getattr(back, attr).detach().cpu().numpy(), z[k]), k


@pytest.mark.gpu
def test_accumulate_view_stats_fused_equals_the_two_reference_updates():
    """train.py:223-227 for one view: the model's one-launch form (gs2m_losses.densification_stats) against
    update_max_radii + add_densification_stats (GM:569-573) on the same state."""
    import types
    P = 5001
    ms = []
    g = torch.Generator().manual_seed(11)
    vg = torch.randn(P, 4, generator=g).cuda()
    vis = (torch.rand(P, generator=g) < 0.7).cuda()
    observe = torch.randint(0, 3, (P,), generator=g, dtype=torch.int32).cuda()
    radii = torch.randint(0, 300, (P,), generator=g, dtype=torch.int32).cuda()
    for fused in (False, True):
        from gs2m_model import GaussianModel
        gm = torch.Generator().manual_seed(0)
        r = lambda *sh: torch.randn(*sh, generator=gm).cuda()
        m = GaussianModel(3, device="cuda")
        m.parameterize((r(P, 3), r(P, 1, 3), r(P, 15, 3), r(P, 3), r(P, 4), r(P, 1), r(P, 3), r(P, 1), r(P, 1)))
        m.xyz_gradient_accum = torch.rand(P, 1, generator=torch.Generator().manual_seed(1)).cuda()
        m.xyz_gradient_accum_abs = torch.rand(P, 1, generator=torch.Generator().manual_seed(2)).cuda()
        m.denom = torch.randint(0, 5, (P, 1), generator=torch.Generator().manual_seed(3)).float().cuda()
        m.max_radii2D = (torch.rand(P, generator=torch.Generator().manual_seed(4)) * 200).cuda()
        m.accumulate_view_stats(types.SimpleNamespace(grad=vg), vis, observe, radii, fused=fused)
        ms.append(m)
    a, b = ms
    assert torch.equal(a.denom, b.denom) and torch.equal(a.max_radii2D, b.max_radii2D)
    assert torch.allclose(a.xyz_gradient_accum, b.xyz_gradient_accum, rtol=3e-7, atol=0)
    assert torch.allclose(a.xyz_gradient_accum_abs, b.xyz_gradient_accum_abs, rtol=3e-7, atol=0)
