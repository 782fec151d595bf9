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
