"""Generates tests/golden/ref_densify.npz: the REFERENCE's own GaussianModel (scene/gaussian_model.py:362-573) driven through
densify_and_prune (clone + split + prune), reset_opacity, reduce_opacity, prune_points, add_densification_stats and
prune_init_points with the Adam-state surgery, every tensor before and after.  Run in the BUILD container only:

    python tests/golden/make_densify_golden.py

The class is imported from /root/reference as it is.  What the image lacks for that import is stood in for HERE, in the
generator only (nothing of this travels, nothing of it is product or oracle code): `plyfile` and `simple_knn._C` are empty
modules (the methods driven here never touch them), and the class's hard-coded `device="cuda"` / `.cuda()` are mapped to the
CPU by wrapping torch's factory functions for the duration of the run.  torch.normal's draws are recorded with their `std`
argument: the test replays them (a generator's stream differs between CPU and GPU), and checks that the product asks for the
same draws.  tests/test_model.py holds gs2m_model.GaussianModel to these vectors on the CPU (bit for bit) and on the GPU."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "albedo", "roughness", "metallic")
ORDER = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic")  # parameterize()'s order


def import_reference_model():
    sys.modules.setdefault("plyfile", types.ModuleType("plyfile"))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = None
    pkg, sub = types.ModuleType("simple_knn"), types.ModuleType("simple_knn._C")
    sub.distCUDA2 = None
    pkg._C = sub
    sys.modules.setdefault("simple_knn", pkg)
    sys.modules.setdefault("simple_knn._C", sub)
    for name in ("zeros", "ones", "tensor", "empty", "full", "rand", "randn", "arange"):
        orig = getattr(torch, name)
        def wrapped(*a, __orig=orig, **k):
            if str(k.get("device", "")) == "cuda":
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(torch, name, wrapped)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, "/root/reference")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_gaussian_model", "/root/reference/scene/gaussian_model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.GaussianModel


class Args:  # arguments/__init__.py: OptimizationParams defaults (tests/golden/ref_defaults.json)
    percent_dense = 0.01
    position_lr_init, position_lr_final, position_lr_delay_mult, position_lr_max_steps = 0.00016, 0.0000016, 0.01, 30000
    feature_lr, opacity_lr, scaling_lr, rotation_lr = 0.0025, 0.025, 0.005, 0.001
    prune_init_points = False


def snapshot(out, tag, m):
    for grp in m.optimizer.param_groups:
        p = grp["params"][0]
        st = m.optimizer.state[p]
        out[f"{tag}/p/{grp['name']}"] = p.detach().numpy().copy()
        out[f"{tag}/m/{grp['name']}"] = st["exp_avg"].numpy().copy()
        out[f"{tag}/v/{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
    # the class's attributes are the optimizer's parameters (the surgery rebinds them)
    for attr, name in (("_xyz", "xyz"), ("_features_dc", "f_dc"), ("_features_rest", "f_rest"), ("_opacity", "opacity"), ("_scaling", "scaling"),
                       ("_rotation", "rotation"), ("_albedo", "albedo"), ("_roughness", "roughness"), ("_metallic", "metallic")):
        assert getattr(m, attr) is [g for g in m.optimizer.param_groups if g["name"] == name][0]["params"][0], (tag, attr)
    out[f"{tag}/accum"] = m.xyz_gradient_accum.numpy().copy()
    out[f"{tag}/accum_abs"] = m.xyz_gradient_accum_abs.numpy().copy()
    out[f"{tag}/denom"] = m.denom.numpy().copy()
    out[f"{tag}/max_radii"] = m.max_radii2D.numpy().copy()


def margin_ok(m, max_grad, max_grad_abs, min_opacity, extent, screen):
    """every comparison the densification makes is decided by more than 1e-4 relative: the GPU's exp / sigmoid / norm may differ in the last bits"""
    rel = lambda a, b: (torch.abs(a - b) / b).min().item() if a.numel() else 1.0
    g = torch.nan_to_num(m.xyz_gradient_accum / m.denom, nan=0.0)
    ga = torch.nan_to_num(m.xyz_gradient_accum_abs / m.denom, nan=0.0)
    big = m.get_scaling.max(dim=1).values
    checks = [rel(torch.norm(g, dim=-1), max_grad), rel(ga.squeeze(), max_grad_abs), rel(big, m.percent_dense * extent),
              rel(big / 1.6, m.percent_dense * extent), rel(big, 0.1 * extent), rel(big / 1.6, 0.1 * extent), rel(m.get_opacity.squeeze(), min_opacity)]
    if screen:
        checks.append(rel(m.max_radii2D, float(screen)))
    return min(checks) > 1e-4


def main():
    GaussianModel = import_reference_model()
    out = {}
    n, extent = 240, 4.0
    seed = 11
    while True:
        g = torch.Generator().manual_seed(seed)
        prm = dict(xyz=torch.randn(n, 3, generator=g), f_dc=torch.randn(n, 1, 3, generator=g), f_rest=torch.randn(n, 15, 3, generator=g),
                   opacity=torch.randn(n, 1, generator=g) * 2.5, scaling=torch.randn(n, 3, generator=g) * 0.8 - 3.0,
                   rotation=torch.randn(n, 4, generator=g), albedo=torch.randn(n, 3, generator=g), roughness=torch.randn(n, 1, generator=g),
                   metallic=torch.randn(n, 1, generator=g))
        m = GaussianModel(3)
        m.spatial_lr_scale = 2.5
        m.parameterize([prm[k].clone() for k in ORDER])
        m.max_radii2D = torch.zeros(n)
        m.training_setup(Args)
        for grp in m.optimizer.param_groups:  # one Adam step: every parameter gets non-trivial moments
            p = grp["params"][0]
            p.grad = torch.randn(p.shape, generator=g)
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        out.clear()
        snapshot(out, "s0", m)

        draws = []
        orig_normal = torch.normal
        def recording_normal(*a, **k):
            r = orig_normal(*a, **k)
            draws.append((k["std"].detach().clone(), r.detach().clone()))
            return r
        torch.normal = recording_normal
        ok = True
        stage = 0
        for rnd, screen in ((0, None), (1, 20)):
            cnt = m.get_xyz.shape[0]
            acc = torch.rand(cnt, 1, generator=g) * 6e-4
            acc_abs = torch.rand(cnt, 1, generator=g) * 2.4e-3
            den = (torch.rand(cnt, 1, generator=g) > 0.1).float() * 3  # some Gaussians never seen: 0 / 0 -> NaN -> 0
            rad = torch.rand(cnt, generator=g) * 40
            m.xyz_gradient_accum, m.xyz_gradient_accum_abs, m.denom, m.max_radii2D = acc.clone(), acc_abs.clone(), den.clone(), rad.clone()
            for k, v in (("accum", acc), ("accum_abs", acc_abs), ("denom", den), ("max_radii", rad)):
                out[f"in{rnd}/{k}"] = v.numpy().copy()
            ok = ok and margin_ok(m, 0.0002, 0.0008, 0.005, extent, screen)
            torch.manual_seed(500 + rnd)
            m.densify_and_prune(0.0002, 0.0008, 0.005, extent, screen)
            stage += 1
            snapshot(out, f"s{stage}", m)  # s1: after round 0; s3: after round 1
            if rnd == 0:
                m.reset_opacity()
                stage += 1
                snapshot(out, f"s{stage}", m)  # s2
        torch.normal = orig_normal
        if ok:
            break
        seed += 1
    out["seed"] = np.array(seed)
    out["args"] = np.array([0.0002, 0.0008, 0.005, extent, Args.percent_dense, 2.5])
    for i, (std, r) in enumerate(draws):
        out[f"normal{i}/std"] = std.numpy().copy()
        out[f"normal{i}/out"] = r.numpy().copy()
    assert len(draws) == 2

    # s4: reduce_opacity (GM:367-370); s5: prune_points with an explicit mask (GM:405-424); s6: add_densification_stats (GM:569-573)
    m.reduce_opacity()
    snapshot(out, "s4", m)
    cnt = m.get_xyz.shape[0]
    mask = torch.rand(cnt, generator=g) < 0.3
    out["in5/mask"] = mask.numpy().copy()
    m.prune_points(mask)
    snapshot(out, "s5", m)
    cnt = m.get_xyz.shape[0]
    vs = torch.zeros(cnt, 4, requires_grad=True)
    vs.grad = torch.randn(cnt, 4, generator=g) * 1e-3
    filt = torch.rand(cnt, generator=g) < 0.6
    out["in6/grad"], out["in6/filter"] = vs.grad.numpy().copy(), filt.numpy().copy()
    m.add_densification_stats(vs, filt)
    m.add_densification_stats(vs, filt)
    snapshot(out, "s6", m)
    # s7: prune_init_points (GM:426-435): the mean / 0.999-quantile rule on a fresh model
    m.prune_init_points()
    snapshot(out, "s7", m)
    assert out["s7/p/xyz"].shape[0] < out["s6/p/xyz"].shape[0]
    np.savez_compressed(os.path.join(HERE, "ref_densify.npz"), **out)
    print("wrote ref_densify.npz: seed", seed, "points per stage", [out[f"s{i}/p/xyz"].shape[0] for i in range(8)])


if __name__ == "__main__":
    main()
