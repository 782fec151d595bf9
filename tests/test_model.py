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
# densification / pruning / opacity reset against a functional restatement of scene/gaussian_model.py:372-567
def _restated_build_rotation(r):  # utils/general_utils.py:76-98
    q = r / torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


class _RestatedModel:
    """The reference's densification written as plain tensor bookkeeping: a dict of parameters, a dict of Adam moments
    per parameter, the three statistics and max_radii2D.  Each step cites the reference lines it follows."""
    NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "albedo", "roughness", "metallic")

    def __init__(self, params, moments, percent_dense):
        self.p = {k: v.clone() for k, v in params.items()}
        self.m = {k: (a.clone(), b.clone()) for k, (a, b) in moments.items()}
        self.percent_dense = percent_dense
        self.dev = params["xyz"].device
        n = self.p["xyz"].shape[0]
        self._reset(n)

    def _reset(self, n):
        z = lambda *sh: torch.zeros(*sh, device=self.dev)
        self.accum, self.accum_abs, self.denom = z(n, 1), z(n, 1), z(n, 1)
        self.max_radii = z(n)

    def _append(self, new):  # cat_tensors_to_optimizer GM:430-455 + densification_postfix GM:459-492
        for k in self.NAMES:
            self.p[k] = torch.cat((self.p[k], new[k]), dim=0)
            a, b = self.m[k]
            self.m[k] = (torch.cat((a, torch.zeros_like(new[k])), dim=0), torch.cat((b, torch.zeros_like(new[k])), dim=0))
        self._reset(self.p["xyz"].shape[0])

    def _prune(self, mask):  # prune_points GM:396-415 (+ _prune_optimizer GM:378-394)
        keep = ~mask
        for k in self.NAMES:
            self.p[k] = self.p[k][keep]
            self.m[k] = (self.m[k][0][keep], self.m[k][1][keep])
        self.accum, self.accum_abs, self.denom = self.accum[keep], self.accum_abs[keep], self.denom[keep]
        self.max_radii = self.max_radii[keep]

    def densify_and_prune(self, max_grad, max_grad_abs, min_opacity, extent, max_screen_size=None):  # GM:538-560
        grads = self.accum / self.denom
        grads[grads.isnan()] = 0.0
        grads_abs = self.accum_abs / self.denom
        grads_abs[grads_abs.isnan()] = 0.0
        # densify_and_clone GM:516-536
        big = torch.max(torch.exp(self.p["scaling"]), dim=1).values
        sel = (torch.norm(grads, dim=-1) >= max_grad) & (big <= self.percent_dense * extent)
        self._append({k: self.p[k][sel] for k in self.NAMES})
        # densify_and_split GM:489-514, N = 2
        N = 2
        n = self.p["xyz"].shape[0]
        padded = torch.zeros(n, device=self.dev)
        padded[:grads_abs.shape[0]] = grads_abs.squeeze()
        act = torch.exp(self.p["scaling"])
        sel = (padded >= max_grad_abs) & (torch.max(act, dim=1).values > self.percent_dense * extent)
        stds = act[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=self.dev), std=stds)
        rots = _restated_build_rotation(self.p["rotation"][sel]).repeat(N, 1, 1)
        new = {k: self.p[k][sel].repeat(N, *([1] * (self.p[k].dim() - 1))) for k in self.NAMES}
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.p["xyz"][sel].repeat(N, 1)
        new["scaling"] = torch.log(act[sel].repeat(N, 1) / (0.8 * N))
        self._append(new)
        self._prune(torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=torch.bool, device=self.dev))))
        # transparent / large GM:549-558 (max_radii2D was just reset by the postfix, as in the reference)
        prune = (torch.sigmoid(self.p["opacity"]) < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii > max_screen_size) | (torch.exp(self.p["scaling"]).max(dim=1).values > 0.1 * extent)
        self._prune(prune)

    def reset_opacity(self):  # GM:362-365 + replace_tensor_to_optimizer GM:372-386
        s = torch.sigmoid(self.p["opacity"])
        v = torch.min(s, torch.ones_like(s) * 0.01)
        self.p["opacity"] = torch.log(v / (1 - v))
        self.m["opacity"] = (torch.zeros_like(self.p["opacity"]), torch.zeros_like(self.p["opacity"]))


@pytest.mark.parametrize("device", DEVICES)
def test_densify_prune_reset_match_the_restated_reference(device):
    """clone + split (torch.normal under a fixed generator) + prune with the Adam-moment surgery, then an opacity reset
    and a second round with max_screen_size: GaussianModel vs the restatement above on the same device, bit for bit.
    On the GPU the optimizer is the product's fused Adam (gs2m_optim.Adam), whose state the surgery edits in training."""
    from gs2m_model import GaussianModel, OptimizationParams
    g = torch.Generator().manual_seed(11)
    n, extent = 400, 4.0
    prm = dict(xyz=torch.randn(n, 3, generator=g), f_dc=torch.randn(n, 1, 3, generator=g), f_rest=torch.randn(n, 15, 3, generator=g),
               opacity=torch.randn(n, 1, generator=g) * 2.5, scaling=torch.randn(n, 3, generator=g) * 0.8 - 3.0,
               rotation=torch.randn(n, 4, generator=g), albedo=torch.randn(n, 3, generator=g), roughness=torch.randn(n, 1, generator=g),
               metallic=torch.randn(n, 1, generator=g))
    model = GaussianModel(3, device)
    model.parameterize([prm[k].clone() for k in ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic")])

    class Opt(OptimizationParams):
        prune_init_points = False
        percent_dense = 0.01
    model.training_setup(Opt, optimizer_cls=torch.optim.Adam if device == "cpu" else None)
    # one optimizer step so that every parameter has non-trivial Adam moments
    for grp in model.optimizer.param_groups:
        p = grp["params"][0]
        p.grad = torch.randn(p.shape, generator=g).to(device)
    model.optimizer.step()
    model.optimizer.zero_grad(set_to_none=True)
    params = {grp["name"]: grp["params"][0].detach().clone() for grp in model.optimizer.param_groups}
    moments = {grp["name"]: (model.optimizer.state[grp["params"][0]]["exp_avg"].clone(),
                             model.optimizer.state[grp["params"][0]]["exp_avg_sq"].clone()) for grp in model.optimizer.param_groups}
    ref = _RestatedModel(params, moments, Opt.percent_dense)

    def check(tag):
        assert model.get_xyz.shape[0] == ref.p["xyz"].shape[0], tag
        for grp in model.optimizer.param_groups:
            k, p = grp["name"], grp["params"][0]
            assert torch.equal(p.detach(), ref.p[k]), (tag, k)
            st = model.optimizer.state.get(p)
            assert st is not None and torch.equal(st["exp_avg"], ref.m[k][0]) and torch.equal(st["exp_avg_sq"], ref.m[k][1]), (tag, k)
        assert torch.equal(model.max_radii2D, ref.max_radii) and torch.equal(model.denom, ref.denom)

    for rnd, screen in ((0, None), (1, 20)):
        m = model.get_xyz.shape[0]
        acc = (torch.rand(m, 1, generator=g) * 6e-4).to(device)
        acc_abs = (torch.rand(m, 1, generator=g) * 2.4e-3).to(device)
        den = ((torch.rand(m, 1, generator=g) > 0.1).float() * 3).to(device)  # some Gaussians never seen: 0 / 0 -> NaN -> 0
        rad = (torch.rand(m, generator=g) * 40).to(device)
        model.xyz_gradient_accum, model.xyz_gradient_accum_abs, model.denom, model.max_radii2D = acc.clone(), acc_abs.clone(), den.clone(), rad.clone()
        ref.accum, ref.accum_abs, ref.denom, ref.max_radii = acc.clone(), acc_abs.clone(), den.clone(), rad.clone()
        torch.manual_seed(500 + rnd)
        model.densify_and_prune(0.0002, 0.0008, 0.005, extent, screen)
        torch.manual_seed(500 + rnd)
        ref.densify_and_prune(0.0002, 0.0008, 0.005, extent, screen)
        check(f"round {rnd}")
        assert model.get_xyz.shape[0] != m
        if rnd == 0:
            model.reset_opacity()
            ref.reset_opacity()
            check("reset")


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
