"""TEST INFRASTRUCTURE ONLY -- never imported by the product (gs-2m_amd/).

numpy (float64) restatement of render-utils' cubemap prefilters (submodules/render-utils/c_src/cubemap.cu): texel
directions (:32-46), pixel_area (:17-30), DiffuseCubemapFwdKernel (:110-140), ndfGGX (:193-198) and
SpecularCubemapFwdKernel (:262-306) as dense weight matrices, so forward = W @ x and backward = W.T @ g.  The bounds
table of the reference is an acceleration structure (every texel inside a box is still tested against the cutoff) and is
not restated.  Parity unpinned: the reference's tests hold no stored vectors for these operators."""
import numpy as np


def texel_dirs(N):
    c = 2.0 * ((np.arange(N) + 0.5) / N) - 1.0
    fy, fx = np.meshgrid(c, c, indexing="ij")
    one = np.ones_like(fx)
    faces = [(one, -fy, -fx), (-one, -fy, fx), (fx, one, fy), (fx, -one, -fy), (fx, -fy, one), (-fx, -fy, -one)]
    d = np.stack([np.stack(f, axis=-1) for f in faces], axis=0).reshape(-1, 3)
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def texel_areas(N):
    if N <= 1:
        return np.ones(6 * N * N)
    H = N // 2
    k = np.abs(np.arange(N) - H)
    a = np.arctan((k + 1) / H) - np.arctan(k / H)
    return np.tile(np.outer(a, a).reshape(-1), 6)  # [y, x] -> dy * dx


def diffuse_matrix(N):
    d = texel_dirs(N)
    return np.clip(d @ d.T, 0.0, 0.999) * texel_areas(N)[None, :] / 3.141592


def specular_matrix(N, roughness, cos_cut, dot_dtype=np.float64, rows=None):
    """Dense weights W[o, t]; `rows`: only these output texels (a (len(rows), 6 N^2) slice, for resolutions whose full
    matrix does not fit)."""
    d = texel_dirs(N)
    do = d if rows is None else d[np.asarray(rows)]
    dots = (do.astype(dot_dtype) @ d.astype(dot_dtype).T).astype(np.float64)
    inside = dots >= cos_cut
    H = do[:, None, :] + d[None, :, :]
    H /= np.maximum(np.linalg.norm(H, axis=-1, keepdims=True), 1e-300)
    vh = np.clip(np.einsum("oc,otc->ot", do, H), 0.0, 1.0)
    a2 = roughness ** 4
    den = (vh * a2 - vh) * vh + 1.0
    W = np.maximum(dots, 0.0) * (a2 / (den * den * np.pi)) * texel_areas(N)[None, :] / 4.0
    return np.where(inside, W, 0.0)
