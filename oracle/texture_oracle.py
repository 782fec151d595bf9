"""TEST INFRASTRUCTURE ONLY -- never imported by the product (gs-2m_amd/).

numpy (float64) restatement of the three `nvdiffrast.torch.texture` modes the reference uses, following
submodules/nvdiffrast/nvdiffrast/common/textureCUDA.cu: face selection and face coordinates (indexCubeMap :99-121),
texel-space coordinates and the four-texel footprint (indexTextureLinear :362-470), continuation of the footprint
across cube edges (wrapCubeMap :47-92 -- restated geometrically: the texel centre is folded around the edge, and the
neighbour texel is found by SEARCH over the neighbouring face, not by the index algebra the kernel uses), the corner
texel as the average of the other three (fetchQuad :590-607), level selection from the bias alone (calculateMipLevel
:575-589) and the blend of the two levels (:776-779).  Parity unpinned in the strict sense (no golden vectors in the
reference; nvdiffrast is CUDA only): besides this restatement the tests rely on properties that hold for any correct
implementation -- continuity across edges and corners, exactness at texel centres, constants, and the adjoint
identity for the backward."""
import numpy as np

_FACE_POINT = {0: lambda X, Y: (1.0, -Y, -X), 1: lambda X, Y: (-1.0, -Y, X), 2: lambda X, Y: (X, 1.0, Y),
               3: lambda X, Y: (X, -1.0, -Y), 4: lambda X, Y: (X, -Y, 1.0), 5: lambda X, Y: (-X, -Y, -1.0)}


def cube_index(d):
    """direction (3,) -> (face, u, v) with u, v in [0, 1], or None for a non-finite result."""
    x, y, z = (float(np.float32(c)) for c in d)
    ax, ay, az = abs(x), abs(y), abs(z)
    if az > max(ax, ay):
        f, c, s, t = 4, z, x, y
    elif ay > ax:
        f, c, s, t = 2, y, x, z
    else:
        f, c, s, t = 0, x, z, y
    if c < 0:
        f += 1
    with np.errstate(all="ignore"):
        m = np.float64(0.5) / np.float64(abs(c))
        u = s * (-m if f in (0, 5) else m) + 0.5
        v = t * (-m if f != 2 else m) + 0.5
    if not (np.isfinite(u) and np.isfinite(v)):
        return None
    return f, min(max(u, 0.0), 1.0), min(max(v, 0.0), 1.0)


def fold(f, ix, iy, w):
    """texel (ix, iy) of face f, possibly one texel outside -> (face, x, y) on the cube, or None at a corner."""
    ox, oy = not 0 <= ix < w, not 0 <= iy < w
    if not ox and not oy:
        return f, ix, iy
    if ox and oy:
        return None
    X, Y = (2 * ix + 1 - w) / w, (2 * iy + 1 - w) / w          # face coordinates of the texel centre, one beyond +-1
    p = list(_FACE_POINT[f](X, Y))
    major = f >> 1
    a = [k for k in range(3) if k != major and abs(p[k]) > 1.0][0]
    over = abs(p[a]) - 1.0
    p[major] -= np.sign(p[major]) * over                       # bend the overshoot around the edge
    p[a] = np.sign(p[a])
    g = 2 * a + (1 if p[a] < 0 else 0)
    best = None                                                # the texel of face g whose centre is that point
    for jx in range(w):
        for jy in range(w):
            q = _FACE_POINT[g]((2 * jx + 1 - w) / w, (2 * jy + 1 - w) / w)
            e = sum((qa - pa) ** 2 for qa, pa in zip(q, p))
            if best is None or e < best[0]:
                best = (e, jx, jy)
    assert best[0] < 1e-18
    return g, best[1], best[2]


def footprint_cube(d, w):
    r = cube_index(d)
    if r is None:
        return None
    f, u, v = r
    u, v = u * w - 0.5, v * w - 0.5
    iu0, iv0 = int(np.floor(u)), int(np.floor(v))
    fu, fv = u - iu0, v - iv0
    tex = [fold(f, iu0 + (k & 1), iv0 + (k >> 1), w) for k in range(4)]
    wt = [(1 - fu) * (1 - fv), fu * (1 - fv), (1 - fu) * fv, fu * fv]
    if any(t is None for t in tex):
        miss = [k for k in range(4) if tex[k] is None][0]
        share = wt[miss] / 3.0
        wt = [w_ + share for w_ in wt]
    return [(t, w_) for t, w_ in zip(tex, wt) if t is not None]


def cube_sample(levels, dirs, bias=None):
    """levels: list of (6, w, w, C) arrays; dirs (n, 3); bias (n,) or None -> (n, C) float64."""
    levels = [np.asarray(l, dtype=np.float64) for l in levels]
    out = np.zeros((len(dirs), levels[0].shape[-1]))
    for i, d in enumerate(dirs):
        parts = [(0, 1.0)]
        if bias is not None:
            fl = min(max(float(np.float32(bias[i])), 0.0), float(len(levels) - 1))
            l0 = int(np.floor(fl))
            parts = [(l0, 1.0)]
            if fl > 0:
                l1 = min(l0 + 1, len(levels) - 1)
                parts = [(l0, 1.0 - (fl - l0)), (l1, fl - l0)]
        for lv, a in parts:
            fp = footprint_cube(d, levels[lv].shape[1])
            if fp is None:
                continue
            for (f, x, y), w_ in fp:
                out[i] += a * w_ * levels[lv][f, y, x]
    return out


def tex2d_clamp_sample(tex, uv):
    tex = np.asarray(tex, dtype=np.float64)
    H, W, _ = tex.shape
    out = np.zeros((len(uv), tex.shape[-1]))
    for i, (u, v) in enumerate(np.asarray(uv, dtype=np.float32).astype(np.float64)):
        u = min(max(u * W - 0.5, 0.0), W - 1.0)
        v = min(max(v * H - 0.5, 0.0), H - 1.0)
        iu0, iv0 = int(np.floor(u)), int(np.floor(v))
        iu1 = iu0 + (0 if u in (0.0, W - 1.0) else 1)
        iv1 = iv0 + (0 if v in (0.0, H - 1.0) else 1)
        fu, fv = u - iu0, v - iv0
        out[i] = ((1 - fu) * (1 - fv) * tex[iv0, iu0] + fu * (1 - fv) * tex[iv0, iu1] + (1 - fu) * fv * tex[iv1, iu0] + fu * fv * tex[iv1, iu1])
    return out
