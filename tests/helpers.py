"""Shared helpers for the parity tests: scene construction, oracle calls, comparisons."""
import numpy as np
import torch

import gs2m_synth as S

# north_star tolerances
ABS_TOL_BUFFERS = 1e-4
REL_TOL_GRADS = 1e-3


def make_scene(P, W, H, seed=0, sh_degree=3, fc=10, scale_lo=0.002, scale_hi=0.02, bg=(0.0, 0.0, 0.0),
               cam=None, behind_frac=0.01):
    cam = cam or S.make_camera(W, H)
    g = S.make_gaussians(P, cam, seed=seed, sh_degree=sh_degree, scale_lo=scale_lo, scale_hi=scale_hi,
                         behind_frac=behind_frac)
    Gc, Gb = S.make_upstream_grads(H, W, seed=seed)
    return dict(cam=cam, g=g, Gc=Gc, Gb=Gb, W=W, H=H, fc=fc, sh_degree=sh_degree,
                bg=torch.tensor(bg, dtype=torch.float32))


def run_oracle(oracle, sc, colors_precomp=None, cov3D_precomp=None, backward=True):
    g, cam = sc["g"], sc["cam"]
    kw = dict(bg=sc["bg"].numpy(), viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(),
              campos=cam["campos"].numpy(), W=sc["W"], H=sc["H"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
              sh_degree=sc["sh_degree"], feature_count=sc["fc"], features=g["features"].numpy())
    if colors_precomp is None:
        kw["shs"] = g["shs"].numpy()
    else:
        kw["colors_precomp"] = colors_precomp.numpy()
    if cov3D_precomp is None:
        kw["scales"] = g["scales"].numpy(); kw["rotations"] = g["rotations"].numpy()
    else:
        kw["cov3D_precomp"] = cov3D_precomp.numpy()
    f = oracle.forward(g["means3D"].numpy(), g["opacities"].numpy(), **kw)
    gr = oracle.backward(f, sc["Gc"].numpy(), sc["Gb"].numpy()) if backward else None
    return f, gr


def settings_for(sc, device):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    cam = sc["cam"]
    return GaussianRasterizationSettings(
        image_height=sc["H"], image_width=sc["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=sc["bg"].to(device), scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(device),
        projmatrix=cam["projmatrix"].to(device), sh_degree=sc["sh_degree"], campos=cam["campos"].to(device),
        prefiltered=False, feature_count=sc["fc"])


def run_hip(sc, device="cuda", colors_precomp=None, cov3D_precomp=None, backward=True):
    """Forward (+ backward) through the drop-in GaussianRasterizer; returns numpy dicts."""
    from diff_gaussian_rasterization import GaussianRasterizer
    g = {k: v.to(device).requires_grad_(True) for k, v in sc["g"].items()}
    P = g["means3D"].shape[0]
    means2D = torch.zeros(P, 4, device=device, requires_grad=True)
    rast = GaussianRasterizer(settings_for(sc, device))
    kw = {}
    if colors_precomp is None:
        kw["shs"] = g["shs"]
    else:
        cp = colors_precomp.to(device).requires_grad_(True)
        kw["colors_precomp"] = cp
    if cov3D_precomp is None:
        kw["scales"] = g["scales"]; kw["rotations"] = g["rotations"]
    else:
        c3 = cov3D_precomp.to(device).requires_grad_(True)
        kw["cov3D_precomp"] = c3
    color, radii, observe, buffer = rast(g["means3D"], means2D, g["opacities"], features=g["features"], **kw)
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), observe=observe.cpu().numpy(),
               buffer=buffer.detach().cpu().numpy())
    grads = None
    if backward:
        loss = (color * sc["Gc"].to(device)).sum() + (buffer * sc["Gb"].to(device)).sum()
        loss.backward()
        z = lambda t: None if t.grad is None else t.grad.cpu().numpy()
        grads = dict(means3D=z(g["means3D"]), means2D=z(means2D), opacities=z(g["opacities"]),
                     features=z(g["features"]))
        if colors_precomp is None:
            grads["shs"] = z(g["shs"])
        else:
            grads["colors"] = z(cp)
        if cov3D_precomp is None:
            grads["scales"] = z(g["scales"]); grads["rotations"] = z(g["rotations"])
        else:
            grads["cov3D"] = z(c3)
    return out, grads


def rel_err(a, b):
    """max |a-b| relative to the largest reference magnitude (scale-aware max-norm error)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def frac_exceeding(a, b, tol):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) > tol).mean()) if a.size else 0.0


def assert_image_close(name, got, ref, tol=ABS_TOL_BUFFERS, scale=None, max_outlier_frac=2e-5):
    """abs tolerance `tol` (times the channel's magnitude scale when given); a threshold flip of one
    alpha ~ 1/255 pair (different exp rounding between CPU and GPU) may move single pixels,
    so a vanishing fraction of outliers is tolerated, bounded in size by 2/255 * scale."""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    s = 1.0 if scale is None else scale
    d = np.abs(got - ref)
    frac = float((d > tol * s).mean())
    assert frac <= max_outlier_frac, f"{name}: {frac:.2e} of pixels differ by more than {tol * s:g} (max {d.max():g})"
    assert d.max() <= 2.0 / 255.0 * s * 4 + tol * s, f"{name}: max abs diff {d.max():g}"


def assert_grad_close(name, got, ref, rel=REL_TOL_GRADS):
    e = rel_err(got, ref)
    assert e <= rel, f"{name}: relative error {e:.3e} > {rel:g}"


# ---------------------------------------------------------------------------------------
# golden fixtures
def scene_from_golden(z):
    cam = dict(W=int(z["W"]), H=int(z["H"]), tanfovx=float(z["tanfovx"]), tanfovy=float(z["tanfovy"]),
               viewmatrix=torch.tensor(z["viewmatrix"]), projmatrix=torch.tensor(z["projmatrix"]),
               campos=torch.tensor(z["campos"]))
    g = {k: torch.tensor(z["in_" + k]) for k in ("means3D", "scales", "rotations", "opacities", "shs", "features")}
    return dict(cam=cam, g=g, Gc=torch.tensor(z["Gc"]), Gb=torch.tensor(z["Gb"]), W=cam["W"], H=cam["H"],
                fc=int(z["fc"]), sh_degree=int(z["sh_degree"]), bg=torch.tensor(z["bg"]))


# ---------------------------------------------------------------------------------------
# oracle-backed stand-in for GaussianRasterizer on CPU tensors (TEST ONLY): lets the very same
# render() code run around the oracle so that pre/post-processing and end-to-end autograd can be
# compared with the device path.
class _OracleRasterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, oracle, st, means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, features):
        n = lambda t: None if t is None else t.detach().cpu().numpy()
        f = oracle.forward(n(means3D), n(opacities), shs=n(shs), colors_precomp=n(colors_precomp), scales=n(scales),
                           rotations=n(rotations), cov3D_precomp=n(cov3D_precomp), features=n(features), bg=n(st.bg),
                           viewmatrix=n(st.viewmatrix), projmatrix=n(st.projmatrix), campos=n(st.campos),
                           W=st.image_width, H=st.image_height, tanfovx=st.tanfovx, tanfovy=st.tanfovy,
                           sh_degree=st.sh_degree, scale_modifier=st.scale_modifier, feature_count=st.feature_count)
        ctx.f, ctx.oracle = f, oracle
        ctx.has = [t is not None for t in (shs, colors_precomp, scales, rotations, cov3D_precomp, features)]
        radii, observe = torch.tensor(f.radii), torch.tensor(f.observe)
        ctx.mark_non_differentiable(radii, observe)
        return torch.tensor(f.color), radii, observe, torch.tensor(f.buffer)

    @staticmethod
    def backward(ctx, gc, gr, go, gb):
        f = ctx.f
        gc = torch.zeros(3, f.H, f.W) if gc is None else gc
        gb = torch.zeros(10, f.H, f.W) if gb is None else gb
        g = ctx.oracle.backward(f, gc.numpy(), gb.numpy())
        t = lambda k, on=True: torch.tensor(g[k]) if on else None
        h = ctx.has
        return (None, None, t("means3D"), t("means2D"), t("opacities"), t("shs", h[0]), t("colors", h[1]),
                t("scales", h[2]), t("rotations", h[3]), t("cov3D", h[4]), t("features", h[5]))


def oracle_rasterizer_class(oracle):
    class OracleRasterizer:
        def __init__(self, raster_settings):
            self.raster_settings = raster_settings

        def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                     cov3D_precomp=None, features=None):
            return _OracleRasterFn.apply(oracle, self.raster_settings, means3D, means2D, opacities, shs, colors_precomp,
                                         scales, rotations, cov3D_precomp, features)
    return OracleRasterizer


def model_from_scene(sc, device, requires_grad=False):
    import gs2m_scene
    g = sc["g"]
    f = g["features"]
    args = [g["means3D"], g["shs"], g["scales"], g["rotations"], g["opacities"],
            f[:, 5:8].clamp(0.02, 0.98), f[:, 8:9].clamp(0.02, 0.98), f[:, 9:10].clamp(0.02, 0.98)]
    pc = gs2m_scene.GaussianParams.from_activated(*[a.clone().to(device) for a in args], active_sh_degree=sc["sh_degree"])
    if requires_grad:
        for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_albedo",
                     "_roughness", "_metallic"):
            setattr(pc, name, getattr(pc, name).detach().clone().requires_grad_(True))
    return pc


def render_pair(oracle, sc, grads=False, **kw):
    """render() on the device (HIP op) and on the CPU (same code, oracle-backed op)."""
    import gaussian_renderer
    import gs2m_scene
    res = []
    for device in ("cuda", "cpu"):
        pc = model_from_scene(sc, device, requires_grad=grads)
        cam = gs2m_scene.Camera(sc["cam"], device)
        saved = gaussian_renderer.GaussianRasterizer
        if device == "cpu":
            gaussian_renderer.GaussianRasterizer = oracle_rasterizer_class(oracle)
        try:
            out = gaussian_renderer.render(cam, pc, gs2m_scene.PipelineParams(), sc["bg"].to(device), **kw)
            if grads:
                loss = (out["render"] * sc["Gc"].to(device)).sum() + (out["depth_map"] * sc["Gb"][1:2].to(device)).sum() \
                    + (out["normal_map"] * sc["Gb"][2:5].to(device)).sum() + (out["albedo_map"] * sc["Gb"][5:8].to(device)).sum()
                loss.backward()
        finally:
            gaussian_renderer.GaussianRasterizer = saved
        o = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
        if grads:
            o["param_grads"] = [p.grad.detach().cpu().numpy() for p in pc.parameters()]
            o["viewspace_grad"] = out["viewspace_points"].grad.detach().cpu().numpy()
        res.append(o)
    return res[0], res[1]
