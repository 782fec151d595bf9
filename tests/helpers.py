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


def run_hip_sums(sc, device="cuda"):
    """Forward + backward through the `_C` functions (the reference's pybind surface), keeping the per-Gaussian SUMS
    the blend backward produces (dL/dmeans2D, dL/dconic, dL/dcolour, dL/dopacity, dL/dfeatures) next to the final
    gradients: the two halves of the backward are checked separately (assert_two_stage)."""
    import diff_gaussian_rasterization as dgr
    g = {k: v.to(device) for k, v in sc["g"].items()}
    st = settings_for(sc, device)
    e = torch.Tensor([])
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, sc["H"], sc["W"], g["shs"], sc["sh_degree"], st.campos, False, sc["fc"])
    res = dgr._C.rasterize_gaussians_backward(
        st.bg, g["means3D"], radii, buffer, e, g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, sc["Gc"].to(device), sc["Gb"].to(device), g["shs"], sc["sh_degree"],
        st.campos, geomB, R, binB, imgB, sc["fc"], return_conics=True)
    names = ("means2D", "colors", "opacities", "means3D", "cov3D", "shs", "scales", "rotations", "features", "conics")
    return {k: v.cpu().numpy() for k, v in zip(names, res)}


def assert_two_stage(oracle, f, gr, hip, floor_frac=1e-5):
    """Parity of the backward in two halves.  (A) the per-Gaussian sums of the blend backward (CR/backward.cu:413-598:
    what the reference accumulates with atomicAdd) against the oracle's, element-wise at north_star's 1e-3.  (B) the
    per-Gaussian chain downstream of them (cov2D / projection / SH / cov3D backward, CR/backward.cu:23-410) against
    the oracle's evaluation of the SAME chain on the HIP path's own sums.  The chain divides by denom^2 of the 2-D
    covariance and multiplies 3x3 matrices whose entries span orders of magnitude: for needle-like Gaussians it
    amplifies last-bit differences of the sums (any two summation orders, the reference's atomics included, differ
    there), so an end-to-end element-wise comparison of dL/dscale, dL/drot measures conditioning, not correctness;
    the split removes that amplification from both halves."""
    for k in ("means2D", "conics", "opacities", "colors", "features"):
        ref = gr[k].reshape(hip[k].shape)
        # dL/dconic is a SIGNED sum of terms s * dx * dy whose magnitudes grow with the square of the distance to the
        # centre: for splats centred far off-screen its small elements sit 1e4 below the terms they are summed from
        assert_grad_close("sum:" + k, hip[k], ref, floor_frac=10 * floor_frac if k == "conics" else floor_frac)
    chain = oracle.backward_pergaussian(f, hip["means2D"], hip["conics"], hip["colors"])
    for k in ("means3D", "shs", "scales", "rotations"):
        # measured: bit-identical (tests/parity_report.py); the bound leaves room for a compiler reassociating one product
        assert_grad_close("chain:" + k, hip[k], chain[k].reshape(hip[k].shape), rel=1e-5, floor_frac=1e-6, max_exceptions=0.0)


def rel_err(a, b):
    """max |a-b| relative to the largest reference magnitude (scale-aware max-norm error)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def frac_exceeding(a, b, tol):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) > tol).mean()) if a.size else 0.0


def pixel_threshold_events(f, x, y, band=1e-4, full=False):
    """Walk pixel (x, y)'s tile list of the oracle forward `f` in the reference's order and arithmetic
    (CR/forward.cu:304-357, fp32, written operation order) and report how close the pixel comes to one of the
    discontinuities of the blend: alpha crossing 1/255 (`:336`), power crossing 0 (`:329`), test_T crossing 1e-4
    (`:339-343`).  Returns the smallest relative distance to a threshold seen before the pixel terminates; a pixel
    whose value may legitimately flip between two correct fp32 implementations (exp() differs by an ulp) has an
    event within `band`."""
    f32 = np.float32
    tile = (y // 16) * f.tiles_x + (x // 16)
    lo, hi = int(f.ranges[tile, 0]), int(f.ranges[tile, 1])
    T = f32(1.0)
    best = np.inf
    last = 0
    for k, gid in enumerate(f.vals_sorted[lo:hi]):
        mx, my = f.means2D[gid]
        A, B, C, op = f.conic_opacity[gid]
        dx = f32(mx - f32(x)); dy = f32(my - f32(y))
        t1 = f32(f32(A * dx) * dx); t2 = f32(f32(C * dy) * dy); t3 = f32(f32(B * dx) * dy)
        power = f32(f32(f32(-0.5) * f32(t1 + t2)) - t3)
        best = min(best, abs(float(power)) / 1e-2)  # power within 1e-6 of 0 counts as an event at band 1e-4
        if power > 0:
            continue
        a_raw = float(op) * float(np.exp(np.float64(power)))
        best = min(best, abs(a_raw * 255.0 - 1.0))
        alpha = min(0.99, a_raw)
        if alpha < 1.0 / 255.0:
            continue
        test_T = float(T) * (1.0 - alpha)
        best = min(best, abs(test_T / 1e-4 - 1.0))
        if f32(T * f32(1.0 - f32(alpha))) < f32(1e-4):
            break
        T = f32(T * f32(1.0 - f32(alpha)))
        last = k + 1
    return (best, float(T), last) if full else best


def tile_walk_events(f, tile, upto, observe=True):
    """pixel_threshold_events for all pixels of `tile` at once, up to and including list position `upto`: per pixel the
    smallest relative distance to a discontinuity of the blend seen so far (alpha at 1/255, power at 0, test_T at 1e-4)
    and (observe=True), at entry `upto` itself, of the pre-update transmittance to 0.5 -- the `observe` condition
    (CR/forward.cu:348-350).
    Same arithmetic and order as the reference's loop (fp32 decisions)."""
    f32 = np.float32
    tx, ty = tile % f.tiles_x, tile // f.tiles_x
    px, py = np.meshgrid(np.arange(tx * 16, tx * 16 + 16), np.arange(ty * 16, ty * 16 + 16))
    live = (px < f.W) & (py < f.H)
    pxf, pyf = px.astype(f32), py.astype(f32)
    lo = int(f.ranges[tile, 0])
    T = np.ones((16, 16), f32)
    ev = np.full((16, 16), np.inf)
    for k in range(upto + 1):
        gid = int(f.vals_sorted[lo + k])
        mx, my = f.means2D[gid]
        A, B, C, op = f.conic_opacity[gid]
        dx = f32(mx) - pxf; dy = f32(my) - pyf
        t1 = (f32(A) * dx) * dx; t2 = (f32(C) * dy) * dy; t3 = (f32(B) * dx) * dy
        power = f32(-0.5) * (t1 + t2) - t3
        with np.errstate(over="ignore", invalid="ignore"):
            ev = np.where(live, np.minimum(ev, np.abs(power.astype(np.float64)) / 1e-2), ev)
            pos = power <= 0
            a_raw = float(op) * np.exp(np.minimum(power, 0).astype(np.float64))
            ev = np.where(live & pos, np.minimum(ev, np.abs(a_raw * 255.0 - 1.0)), ev)
            alpha = np.minimum(0.99, a_raw)
            keep = pos & (alpha >= 1.0 / 255.0)
            ev = np.where(live & keep, np.minimum(ev, np.abs(T.astype(np.float64) * (1.0 - alpha) / 1e-4 - 1.0)), ev)
            if k == upto and observe:
                ev = np.where(live, np.minimum(ev, np.abs(T.astype(np.float64) / 0.5 - 1.0)), ev)
            tnew = T * (f32(1.0) - alpha.astype(f32))
        fin = keep & (tnew < f32(1e-4))
        contrib = live & keep & ~fin
        T = np.where(contrib, tnew, T)
        live = live & ~fin
    return ev


def observe_event_cost(f, gid):
    """list entries a proof for Gaussian `gid` has to walk (its instances x their positions in the tile lists)"""
    idx = np.nonzero(f.vals_sorted == gid)[0]
    tiles = (f.keys_sorted[idx] >> np.uint64(32)).astype(np.int64)
    return int((idx - f.ranges[tiles, 0].astype(np.int64) + 1).sum())


def observe_event(f, gid, observe=True, band=None):
    """closest approach of any pixel of Gaussian `gid` to an event that may change observe[gid] (observe=True) or the
    set of pixels it contributes to (observe=False) between two correct fp32 implementations (stops at the first
    instance with an event within `band`)"""
    best = np.inf
    for idx in np.nonzero(f.vals_sorted == gid)[0]:
        tile = int(f.keys_sorted[idx] >> np.uint64(32))
        best = min(best, float(tile_walk_events(f, tile, int(idx) - int(f.ranges[tile, 0]), observe).min()))
        if band is not None and best <= band:
            break
    return best


PROOF_BUDGET = 60_000  # ~10 s of numpy per assertion
def assert_observe_close(got, f, band=1e-4, max_proofs=40):
    """`observe` (CR/forward.cu:348-350: pixels a Gaussian contributes to while T > 0.5) is an integer: equal, except
    for a vanishing number of Gaussians (<= P / 2000), each off by at most two pixels, and each of those must be shown
    to own a pixel that sits on a threshold of the blend (alpha at 1/255, test_T at 1e-4, power at 0) or has T within
    `band` of 0.5 at the Gaussian's own entry -- computed from the oracle's state, as for image outliers."""
    d = np.abs(np.asarray(got).astype(np.int64) - f.observe.astype(np.int64))
    bad = np.nonzero(d)[0]
    assert len(bad) <= max(1, f.P // 2000), f"observe: {len(bad)} Gaussians differ"
    assert d.max(initial=0) <= 2, f"observe differs by {d.max()} pixels on one Gaussian"
    budget = PROOF_BUDGET  # list entries walked in Python: the proofs stop when it is spent (the count bounds above hold for all)
    for gid in bad[:max_proofs]:
        budget -= observe_event_cost(f, int(gid))
        if budget < 0:
            break
        ev = observe_event(f, int(gid), band=band)
        assert ev <= band, f"observe[{gid}] differs by {d[gid]} but no pixel of the Gaussian is at a threshold (closest {ev:.3e})"


def assert_image_close(name, got, ref, tol=ABS_TOL_BUFFERS, scale=None, max_outlier_frac=2e-5, oracle_fwd=None,
                       band=1e-4):
    """abs tolerance `tol` (times the channel's magnitude scale when given).  Pixels outside the tolerance are only
    accepted when they are few AND (with `oracle_fwd`, the OracleForward that produced `ref`) each of them provably
    sits on a discontinuity of the blend: some list entry has alpha within `band` (relative) of 1/255, or test_T
    within `band` of 1e-4, computed from the oracle's own state -- two correct fp32 implementations whose exp()
    differ by an ulp may then take different branches.  Anything else fails."""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    s = 1.0 if scale is None else scale
    d = np.abs(got - ref)
    bad = d > tol * s
    frac = float(bad.mean())
    # (with the proof below, never fewer than 2 pixels: one proven threshold pixel of a 97 x 255 image is 4e-5 of it)
    allowed = max(max_outlier_frac, 2.0 / max(bad.size, 1)) if oracle_fwd is not None else max_outlier_frac
    assert frac <= allowed, f"{name}: {frac:.2e} of pixels differ by more than {tol * s:g} (max {d.max():g})"
    assert d.max() <= 2.0 / 255.0 * s * 4 + tol * s, f"{name}: max abs diff {d.max():g}"
    if oracle_fwd is not None and bad.any():
        pix = bad.reshape(-1, bad.shape[-2], bad.shape[-1]).any(0)
        for y, x in zip(*np.nonzero(pix)):
            ev = pixel_threshold_events(oracle_fwd, int(x), int(y), band)
            assert ev <= band, (f"{name}: pixel ({x},{y}) differs by {d.reshape(-1, *pix.shape)[:, y, x].max():g} but is "
                                f"not at a threshold (closest event {ev:.3e} > {band:g})")


def assert_n_contrib_close(nc, f, band=1e-4, max_frac=1e-4, max_proofs=64):
    """n_contrib (the position of a pixel's last contributor, CR/forward.cu:352-357) is an integer: equal except on a
    vanishing number of pixels (<= 1e-4 of them), each of which must be shown -- from the oracle's own state, as for image
    outliers -- to sit on a threshold of the blend (an entry with alpha within `band` of 1/255, power within 1e-6 of 0 or
    test_T within `band` of 1e-4), where two correct fp32 implementations may take different branches."""
    bad = np.argwhere(np.asarray(nc) != f.n_contrib)
    assert len(bad) <= max_frac * f.n_contrib.size, f"n_contrib differs on {len(bad)} pixels"
    for y, x in bad[:max_proofs]:
        ev = pixel_threshold_events(f, int(x), int(y), band)
        assert ev <= band, f"n_contrib differs at pixel ({x},{y}) ({nc[y, x]} vs {f.n_contrib[y, x]}) but the pixel is not at a threshold (closest event {ev:.3e})"


def grad_stats(got, ref, rel=REL_TOL_GRADS, floor_frac=1e-5):
    """Element-wise comparison: an element passes when |a-b| <= rel*|b| + floor, floor = floor_frac * rms(b) over
    b's non-zero elements (the absolute error an fp32 sum of that tensor's typical terms carries; the oracle
    accumulates in double).  Returns (fraction of elements failing, worst |a-b| / max|b|, floor)."""
    a = np.asarray(got, dtype=np.float64).ravel(); b = np.asarray(ref, dtype=np.float64).ravel()
    if a.size == 0:
        return 0.0, 0.0, 0.0
    nz = b[b != 0]
    rms = float(np.sqrt(np.mean(nz * nz))) if nz.size else 0.0
    floor = floor_frac * rms
    d = np.abs(a - b)
    fail = d > rel * np.abs(b) + floor
    return float(fail.mean()), float(d.max() / (np.abs(b).max() + 1e-30)), floor


# Gaussians whose footprint holds an alpha ~ 1/255 pixel that flips between the two implementations change by that
# pixel's (small) term; such elements may exceed the element-wise bound, but they are few and small:
MAX_GRAD_EXCEPTIONS = 2e-4  # fraction of a tensor's elements
# end to end, dL/dscale and dL/drot additionally carry the conditioning of the cov2D -> cov3D -> (S, R) chain (see
# assert_two_stage, which checks the two halves without it): measured <= 1.7e-3 of the elements on the needle scene (small
# scenes only: the full-size comparisons with the reference build ask every exception for its proof, helpers.check_full_size)
MAX_GRAD_EXCEPTIONS_CHAIN = 2e-3
def assert_grad_close(name, got, ref, rel=REL_TOL_GRADS, max_exceptions=None, floor_frac=None):
    """north_star: 1e-3 relative on gradients, ELEMENT-WISE (`|a-b| <= 1e-3 |b| + floor`, see grad_stats), with a
    counted exception set (<= `max_exceptions` of the elements) that is itself bounded in the max norm
    (every element within 1e-3 of the tensor's largest magnitude).  Defaults: floor 1e-5 of the tensor's rms and
    2e-4 exceptions; for the tensors at the end of the covariance chain (scales, rotations, cov3D) compared end to
    end, floor 1e-4 and MAX_GRAD_EXCEPTIONS_CHAIN."""
    chain = name in ("scales", "rotations", "cov3D", "scaling", "rotation")
    if max_exceptions is None:
        max_exceptions = MAX_GRAD_EXCEPTIONS_CHAIN if chain else MAX_GRAD_EXCEPTIONS
    if floor_frac is None:
        floor_frac = 1e-4 if chain else 1e-5
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.all(np.isfinite(got)), f"{name}: non-finite gradient"
    frac, worst, floor = grad_stats(got, ref, rel, floor_frac)
    allowed = max(max_exceptions, 2.0 / max(got.size, 1)) if max_exceptions > 0 else 0.0  # never fewer than 2 elements
    assert frac <= allowed, f"{name}: {frac:.3e} of the elements are outside {rel:g}*|ref| + {floor:.3g}"
    assert worst <= rel, f"{name}: max-norm relative error {worst:.3e} > {rel:g}"


ROW_FLOOR_FRAC = 1e-6  # see assert_sum_close
SUM_GROSS_MAX = 5e-3  # tensor-wide max-norm bound on blend sums whose exceptions are PROVEN threshold events (worst seen: ref_special_sizes 2.0e-3, a C4
                      # view 1.15e-3; round 5 had 2e-2 here.  Relative to a small Gaussian's OWN row a threshold pixel moves far more -- 2e-1 at 2M / 4K,
                      # for the reference build against the oracle too: profiles/r06_error_tail.md -- but not relative to the tensor's largest element)


def assert_sum_close(name, got, ref, f, rel=REL_TOL_GRADS, floor_frac=1e-5, max_proofs=48, budget=MAX_GRAD_EXCEPTIONS):
    """assert_grad_close for a per-Gaussian blend SUM (what the reference accumulates with atomicAdd), with the proof image
    outliers and observe mismatches get when the counted-exception budget does not cover the rows outside the bound: a
    pixel that one implementation takes and the other does not (alpha * 255 = 1.00000 at a contributor: the two exp() differ
    in the last bit) changes that pixel's colour and with it the sums of EVERY Gaussian blended there, the ~100 behind it
    included -- sweep case 89 of tests/ref_report.py: 11 elements of dL/dcolour, 2e-4 of them allowed.  Every Gaussian
    that owns an element outside the bound must then have, in one of its tiles, a pixel within CHAIN_EVENT_BAND of a
    threshold of the blend at or in front of its own list entry, computed from `f`'s state (the walk of observe_event);
    beyond `rel` in the max norm the proofs are asked for whatever the budget, SUM_GROSS_MAX bounds the proven rows.  `budget`: the counted-exception fraction below which no proof is asked for; 0 = north_star's
    "1e-3 relative, full stop": EVERY row with an element outside the bound needs its proof (the full-size comparisons with the
    reference build: a handful of Gaussians per million)."""
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape and np.all(np.isfinite(got)), name
    frac, worst, floor = grad_stats(got, ref, rel, floor_frac)
    # An element of a Gaussian's row that CANCELS (one colour channel of a splat whose other channels sum to ~10: 5.8e-4 +- 3e-6 in
    # fp32, whichever implementation adds it up -- the reference build's own value moves by that much with its atomics' order) is held
    # to ROW_FLOOR_FRAC of the row's largest element (~16 ulp), not to 1e-3 of itself.
    a = got.reshape(got.shape[0], -1).astype(np.float64); b = ref.reshape(ref.shape[0], -1).astype(np.float64)
    bound = rel * np.abs(b) + np.maximum(floor, ROW_FLOOR_FRAC * np.abs(b).max(1, keepdims=True))
    outside = np.abs(a - b) > bound
    frac = float(outside.mean())
    # A threshold pixel can sit on the Gaussian that owns the tensor's LARGEST element (1 run in 8 of the C4 training view: 1.15e-3 in the
    # max norm): beyond `rel` in the max norm is not a failure by itself, it sends EVERY row outside the element-wise bound to the proof
    # below, whatever the budget; a gross bound stays (a wrong kernel is off by O(1)).
    assert worst <= SUM_GROSS_MAX, f"{name}: max-norm relative error {worst:.3e} > {SUM_GROSS_MAX:g}"
    if worst <= rel and frac <= (max(budget, 2.0 / max(got.size, 1)) if budget > 0 else 0.0):
        return
    rows = np.nonzero(outside.any(1))[0]
    assert len(rows) <= max_proofs, f"{name}: {frac:.3e} of the elements are outside {rel:g}*|ref| + {floor:.3g} ({len(rows)} Gaussians: more than a proof is attempted for)"
    budget = 5 * PROOF_BUDGET
    for r in rows:
        budget -= observe_event_cost(f, int(r))
        assert budget >= 0, f"{name}: {frac:.3e} of the elements outside the bound, proof budget exhausted"
        ev = observe_event(f, int(r), observe=False, band=CHAIN_EVENT_BAND)
        assert ev <= CHAIN_EVENT_BAND, (f"{name}: {frac:.3e} of the elements are outside {rel:g}*|ref| + {floor:.3g} and Gaussian {int(r)} has no "
                                       f"threshold pixel at or in front of it (closest event {ev:.3e})")


def sweep_scene(case):
    """the random scene of sweep case `case` -> (scene, reference-binning flag, tag): 1 ... 120 000 Gaussians on 16 x 16 ...
    1280 x 720 images, every feature count and SH degree, scales over three decades, culled fractions, both binning modes"""
    import random
    rng = random.Random(77000 + case)
    P = rng.choice([1, 3, 50, 400, 2000, 8000, 30000, 120000])
    W, H = rng.choice([(16, 16), (31, 47), (64, 48), (130, 70), (320, 200), (333, 201), (640, 360), (97, 255), (1280, 720)])
    fc = rng.choice([0, 1, 3, 5, 8, 9, 10])
    deg = rng.choice([0, 1, 2, 3, 3])
    lo = rng.choice([0.0005, 0.005, 0.02])
    hi = max(rng.choice([0.03, 0.1, 0.5, 1.2]), 2 * lo)
    if P >= 30000:
        hi = min(hi, 0.1)
    seed = rng.randrange(1 << 30)
    refbin = rng.choice([False, True])
    sc = make_scene(P, W, H, seed=seed, fc=fc, sh_degree=deg, scale_lo=lo, scale_hi=hi, bg=(rng.random(), rng.random(), rng.random()),
                    behind_frac=rng.choice([0.0, 0.01, 0.3]))
    if rng.random() < 0.3:
        sc["g"]["opacities"] = torch.clamp(sc["g"]["opacities"] * 2.5, max=0.999)
    return sc, refbin, f"case {case}: P={P} {W}x{H} fc={fc} deg={deg} scales=[{lo},{hi}] seed={seed} refbin={refbin}"


def check_sweep_case(reference, case, precomputed=False):
    """One sweep case: the HIP path (through the drop-in op) against the reference build `reference` (oracle/_ref): radii exact,
    observe and images with threshold proofs, blend sums and the well-conditioned gradients element-wise.  Raises
    AssertionError; returns the case's tag."""
    import gs2m_native
    sc, refbin, tag = sweep_scene(case)
    kw = {}
    if precomputed:  # precomputed colours and 3-D covariances instead of SH + scale / rotation
        import gs2m_scene
        P = sc["g"]["means3D"].shape[0]
        prm = gs2m_scene.GaussianParams.from_activated(sc["g"]["means3D"], sc["g"]["shs"], sc["g"]["scales"], sc["g"]["rotations"], sc["g"]["opacities"],
                                                       torch.full((P, 3), 0.5), torch.full((P, 1), 0.5), torch.full((P, 1), 0.5))
        kw = dict(colors_precomp=torch.rand(P, 3, generator=torch.Generator().manual_seed(case)), cov3D_precomp=prm.get_covariance().contiguous())
        tag += " precomputed"
    check_scene_against(reference, sc, tag, refbin=refbin, **kw)
    return tag


def check_scene_against(reference, sc, tag="", refbin=False, **kw):
    """The HIP path (through the drop-in op) against the reference build `reference` (oracle/_ref) on scene `sc`: radii exact,
    observe and images with threshold proofs, blend sums and the well-conditioned gradients element-wise (every element outside
    1e-3 relative beyond the counted budget needs its proof).  Raises AssertionError."""
    import gs2m_native
    r, rg = run_oracle(reference, sc, **kw)
    try:
        gs2m_native.set_reference_binning(refbin)
        out, g = run_hip(sc, **kw)
        sums = run_hip_sums(sc) if not kw else None
    finally:
        gs2m_native.set_reference_binning(False)
    assert np.array_equal(out["radii"], r.radii), "radii " + tag
    assert_observe_close(out["observe"], r)
    assert_image_close("color " + tag, out["color"], r.color, oracle_fwd=r)
    for ch in range(10):
        assert_image_close(f"buffer[{ch}] " + tag, out["buffer"][ch], r.buffer[ch], scale=max(1.0, float(np.abs(r.buffer[ch]).max())), oracle_fwd=r)
    if sums is not None:
        for k in ("means2D", "conics", "opacities", "colors", "features"):
            assert_sum_close("sum:" + k + " " + tag, sums[k], rg[k].reshape(sums[k].shape), r, floor_frac=1e-4 if k == "conics" else 1e-5)
    for k in ("shs", "opacities", "features", "means2D", "colors"):
        if k in g and g[k] is not None:
            assert_sum_close(k + " " + tag, g[k], rg[k], r)
    return r, rg, out, g, sums


def check_full_size(reference, sc, refbin, tag=""):
    """The HIP path against the reference build at a BASELINE configuration's full size with NO counted-exception budget: radii
    exact; observe and every image outlier with its threshold proof; every blend sum and every gradient element-wise at 1e-3
    relative, and every Gaussian owning an element outside that bound proven to sit on a threshold of the blend (sums and the
    well-conditioned tensors) or to be needle-like / measurably amplified / on a threshold (dL/dscale, dL/drot, dL/dmean3D)."""
    import gs2m_native
    r, rg = run_oracle(reference, sc)
    try:
        gs2m_native.set_reference_binning(refbin)
        out, g = run_hip(sc)
        sums = run_hip_sums(sc)
    finally:
        gs2m_native.set_reference_binning(False)
    assert np.array_equal(out["radii"], r.radii), "radii " + tag
    assert_observe_close(out["observe"], r)
    assert_image_close("color " + tag, out["color"], r.color, oracle_fwd=r)
    for ch in range(10):
        assert_image_close(f"buffer[{ch}] " + tag, out["buffer"][ch], r.buffer[ch], scale=max(1.0, float(np.abs(r.buffer[ch]).max())), oracle_fwd=r)
    for k in ("means2D", "conics", "opacities", "colors", "features"):
        assert_sum_close("sum:" + k + " " + tag, sums[k], rg[k].reshape(sums[k].shape), r, floor_frac=1e-4 if k == "conics" else 1e-5, budget=0.0)
    for k in ("shs", "opacities", "features", "means2D"):
        assert_sum_close(k + " " + tag, g[k], rg[k], r, budget=0.0)
    assert_chain_exceptions_conditioned(r, g, rg, sums=sums, names=("scales", "rotations", "means3D"), tag=tag)
    return r


def cov2d_anisotropy(f):
    """rho = det / (a c) of the 2-D covariance the backward differentiates through (the forward's, plus 0.3 on the
    diagonal, CR/backward.cu:205-207), recovered from the oracle's conic: 1 for a round splat, -> 0 for a needle.  The
    chain cov2D -> conic divides by det^2 (backward.cu:209-219): its three terms cancel to within rho of their size."""
    A, B, C = (f.conic_opacity[:, k].astype(np.float64) for k in range(3))
    with np.errstate(all="ignore"):
        det = A * C - B * B
        a, b, c = C / det + 0.3, -B / det, A / det + 0.3
        rho = (a * c - b * b) / (a * c)
    return np.where(np.isfinite(rho), rho, 1.0)


CHAIN_RHO_MAX = 0.05  # exceptions of the end-to-end dL/dscale, dL/drot check must be at least this needle-like ...
CHAIN_AMP_MIN = 30.0  # ... or sit where the chain amplifies the difference of the blend sums at least this much ...
CHAIN_EVENT_BAND = 3e-4  # ... or own a pixel this close (relative) to a threshold of the blend
CHAIN_GROSS_MAX = 5e-2  # max-norm bound on PROVEN ill-conditioned rows (worst of the 400-scene sweep: 2.7e-2; a wrong kernel is off by O(1))


def assert_chain_exceptions_conditioned(f, g, gr, sums=None, names=("scales", "rotations"), tag="", rel=REL_TOL_GRADS, floor_frac=1e-4):
    """End to end, dL/dscale and dL/drot element-wise at north_star's 1e-3 (floor: 1e-4 of the tensor's rms), max-norm
    error <= 1e-3 -- and every Gaussian that owns an element outside the element-wise bound must be ILL-CONDITIONED in
    a stated sense: either rho = det(cov2D) / (a c) <= CHAIN_RHO_MAX (a needle: measured on the random sweep,
    tests/chain_exceptions.py, every exception has rho <= 0.024, the 1-3 % most needle-like Gaussians of its scene), or,
    with `sums` (the HIP path's own per-Gaussian blend sums, run_hip_sums), the chain provably AMPLIFIES there: the
    relative difference of the Gaussian's outputs is at least CHAIN_AMP_MIN times the relative difference of its
    dL/dconic, dL/dmean2D sums (whose own element-wise check is assert_two_stage's half A; the median Gaussian
    amplifies 2-3 times).  Since the chain is bit-identical on equal sums (half B), such an exception measures the
    conditioning of CR/backward.cu:153-347 at that Gaussian, not an error of either implementation.
    A third kind exists at the full sizes (1-3 Gaussians per million): a well-conditioned Gaussian one of whose pixels
    sits ON a threshold of the blend (alpha at 1/255, test_T at 1e-4) and is taken by one implementation and not by the
    other -- its sums then differ by that pixel's term.  It is accepted only with the same proof image outliers and
    observe mismatches get: an event within CHAIN_EVENT_BAND (relative) computed from the oracle's state (observe_event;
    measured on the 1M scene: 1.9e-7, 3.7e-5, 1.2e-4 -- the transmittance in front of a test_T ~ 1e-4 decision is a
    product of hundreds of factors (1 - alpha) whose rounding differs between two exp() implementations; about 2 % of
    the Gaussians own a pixel that close to a threshold, the three exceptions are among them)."""
    rho = cov2d_anisotropy(f)
    amp_in = None
    if sums is not None:
        so = np.concatenate([np.asarray(gr["conics"]).reshape(-1, 4)[:, [0, 1, 3]], np.asarray(gr["means2D"])[:, :2]], 1).astype(np.float64)
        sh = np.concatenate([np.asarray(sums["conics"]).reshape(-1, 4)[:, [0, 1, 3]], np.asarray(sums["means2D"])[:, :2]], 1).astype(np.float64)
        amp_in = np.linalg.norm(sh - so, axis=1) / (np.linalg.norm(so, axis=1) + 1e-300)
    for k in names:
        got, ref = np.asarray(g[k], dtype=np.float64), np.asarray(gr[k], dtype=np.float64)
        assert got.shape == ref.shape and np.all(np.isfinite(got)), (k, tag)
        nz = ref[ref != 0]
        floor = floor_frac * (float(np.sqrt(np.mean(nz * nz))) if nz.size else 0.0)
        d = np.abs(got - ref)
        rows = np.nonzero((d > rel * np.abs(ref) + floor).any(1))[0]
        # Gross-error guard in the max norm.  Rows inside the element-wise bound meet 1e-3 of the largest element by construction;
        # the rows outside it must each carry a proof below, and for THOSE the end-to-end difference measures the conditioning of
        # the chain at that Gaussian (screen-filling needles are often the largest entries of the tensor): measured on the
        # 400-scene sweep 1.06e-3 ... 2.7e-2 of the largest element, where round 3's flat 1e-3 failed 17 scenes whose two-stage
        # check -- the rigorous half: sums element-wise, chain bit-identical on equal sums -- passes.  CHAIN_GROSS_MAX bounds them.
        peak = np.abs(ref).max(initial=0.0) + 1e-30
        assert d.max(initial=0.0) <= CHAIN_GROSS_MAX * peak, f"{k} {tag}: max-norm relative error {d.max() / peak:.3e}"
        clean = np.ones(d.shape[0], bool)
        clean[rows] = False  # rows WITHOUT an exception keep north_star's bound in the max norm too (CHAIN_GROSS_MAX is for proven rows only)
        assert d[clean].max(initial=0.0) <= rel * peak + floor, f"{k} {tag}: a row inside the element-wise bound exceeds {rel:g} of the largest element"
        ok = rho[rows] <= CHAIN_RHO_MAX
        if amp_in is not None and len(rows):
            with np.errstate(all="ignore"):
                amp = (np.linalg.norm(d[rows], axis=1) / (np.linalg.norm(ref[rows], axis=1) + 1e-300)) / amp_in[rows]
            ok = ok | (amp >= CHAIN_AMP_MIN)
        else:
            amp = np.full(len(rows), np.nan)
        budget = 5 * PROOF_BUDGET
        for q, r in enumerate(rows):
            if not ok[q]:
                budget -= observe_event_cost(f, int(r))
                if budget < 0:
                    break
                if observe_event(f, int(r), observe=False, band=CHAIN_EVENT_BAND) <= CHAIN_EVENT_BAND:
                    ok[q] = True
        bad = rows[~ok]
        assert len(bad) == 0, (f"{k} {tag}: {len(bad)} of {len(rows)} Gaussians outside the element-wise bound are neither needle-like nor amplified nor on a threshold: "
                               + "; ".join(f"gid {r} rho {rho[r]:.3g} amp {amp[list(rows).index(r)]:.3g} radius {f.radii[r]} |ref| {np.abs(ref[r]).max():.3g} d {d[r].max():.3g} floor {floor:.3g}" for r in bad[:5]))
        # (every one of them carries a proof above; the count is a sanity bound: scenes with scales over three decades up to
        # screen-filling splats -- where needles are the rule -- reach 2.3 % on the 400-scene sweep (7 of 300); never fewer than 8)
        assert len(rows) <= max(8, int(3e-2 * got.shape[0])), (k, tag, len(rows))


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
