"""ctypes front-end of oracle/_ref/libgs2m_ref.so: the REFERENCE's own rasterizer kernels (cuda_rasterizer/*.cu passed
through hipify-perl by oracle/ref_build/Makefile in the build container, compiled for gfx950) behind this repository's
C shim (oracle/ref_build/ref_shim.hip).

TEST INFRASTRUCTURE ONLY: only tests/ (its test modules and report scripts) and __graft_entry__ import this module; the product
package never does.  Needs a GPU.  `forward` / `backward` take and return the same things as oracle/oracle.py's, so the two can
stand in for each other in tests/helpers.py (numpy in, numpy out; arguments as rasterize_points.cu:28-129, 131-218
hands them to CudaRasterizer::Rasterizer)."""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libgs2m_ref.so")
NUM_FEATURES = 10
_lib = None


def available():
    return os.path.exists(LIB_PATH) and torch.cuda.is_available()


def build():
    """Only where /root/reference exists (the build container): oracle/ref_build/Makefile."""
    import subprocess
    subprocess.check_call(["make", "-j2", "-C", os.path.join(_HERE, "ref_build")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH


def use_library(path):
    """tests/ref_report.py's timing leg: another build of the same sources (e.g. libgs2m_ref_fast.so, hipcc's default contraction)"""
    global _lib, LIB_PATH
    _lib, LIB_PATH = None, path


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.gs2m_ref_create.restype = C.c_void_p
        L.gs2m_ref_destroy.argtypes = [C.c_void_p]
        p, i, f = C.c_void_p, C.c_int, C.c_float
        L.gs2m_ref_forward.restype = i
        L.gs2m_ref_forward.argtypes = [p, i, i, i, p, i, i, p, p, p, p, p, f, p, p, p, p, p, p, f, f, i, i, p, p, p, p]
        L.gs2m_ref_backward.restype = i
        L.gs2m_ref_backward.argtypes = [p, i, i, i, i, p, i, i, p, p, p, p, f, p, p, p, p, p, p, f, f, p, p, i, p, p] + [p] * 10
        L.gs2m_ref_state.restype = i
        L.gs2m_ref_state.argtypes = [p] * 15
        L.gs2m_ref_mark_visible.restype = i
        L.gs2m_ref_mark_visible.argtypes = [i, p, p, p, p]
        L.gs2m_ref_knn.restype = i
        L.gs2m_ref_knn.argtypes = [i, p, p]
        for name, sig in (("diffuse_cubemap_fwd", [i, p, p]), ("diffuse_cubemap_bwd", [i, p, p, p]), ("specular_bounds", [i, f, p]),
                          ("specular_cubemap_fwd", [i, p, p, f, f, p]), ("specular_cubemap_bwd", [i, p, p, p, f, f, p])):
            fn = getattr(L, "gs2m_ref_" + name)
            fn.restype, fn.argtypes = i, sig
        _lib = L
    return _lib


SSIM_LIB_PATH = os.path.join(_HERE, "_ref", "libgs2m_ref_ssim.so")
_ssim_lib = None


def ssim_available():
    return os.path.exists(SSIM_LIB_PATH) and torch.cuda.is_available()


def _ssim():
    global _ssim_lib
    if _ssim_lib is None:
        L = C.CDLL(SSIM_LIB_PATH)
        p, i, f = C.c_void_p, C.c_int, C.c_float
        L.gs2m_ref_ssim_forward.restype = i
        L.gs2m_ref_ssim_forward.argtypes = [i, i, i, i, f, f, p, p, p, p, p, p]
        L.gs2m_ref_ssim_backward.restype = i
        L.gs2m_ref_ssim_backward.argtypes = [i, i, i, i, f, f, p, p, p, p, p, p, p]
        _ssim_lib = L
    return _ssim_lib


def fusedssim(C1, C2, img1, img2, train):
    """fused-ssim's extension function of the same name (ssim.cu:368-404): (B, CH, H, W) CUDA tensors ->
    (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12), the last three empty when `train` is false"""
    a, b = _dev(img1), _dev(img2)
    B, CH, H, W = a.shape
    out = [torch.empty_like(a) for _ in range(4 if train else 1)]
    torch.cuda.synchronize()
    ptrs = [_ptr(t) for t in out] + [None] * (4 - len(out))
    if _ssim().gs2m_ref_ssim_forward(B, CH, H, W, C1, C2, _ptr(a), _ptr(b), *ptrs) != 0:
        raise RuntimeError("reference fusedssim failed")
    return tuple(out) + tuple(torch.empty(0) for _ in range(4 - len(out)))


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    a, b, g = _dev(img1), _dev(img2), _dev(dL_dmap)
    B, CH, H, W = a.shape
    out = torch.empty_like(a)
    torch.cuda.synchronize()
    if _ssim().gs2m_ref_ssim_backward(B, CH, H, W, C1, C2, _ptr(a), _ptr(b), _ptr(g), _ptr(_dev(dm_dmu1)), _ptr(_dev(dm_dsigma1_sq)),
                                      _ptr(_dev(dm_dsigma12)), _ptr(out)) != 0:
        raise RuntimeError("reference fusedssim_backward failed")
    return out


def _dev(a, dtype=torch.float32):
    if a is None:
        return None
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device="cuda", dtype=dtype).contiguous()


def _ptr(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


class ReferenceForward:
    """Result of one forward of the reference build: API outputs plus the reference's internal arrays, named as
    oracle.OracleForward names them."""

    def __init__(self, handle, out):
        self._h = handle
        self.__dict__.update(out)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib is not None:
                _lib.gs2m_ref_destroy(self._h)
        except Exception:
            pass
        self._h = None


def forward(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, features=None,
            bg, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree=0, scale_modifier=1.0, prefiltered=False,
            feature_count=0, state=True):
    L = lib()
    d = dict(means3D=_dev(means3D), opacities=_dev(opacities), shs=_dev(shs), colors_precomp=_dev(colors_precomp), scales=_dev(scales),
             rotations=_dev(rotations), cov3D_precomp=_dev(cov3D_precomp), features=_dev(features), bg=_dev(bg),
             viewmatrix=_dev(viewmatrix), projmatrix=_dev(projmatrix), campos=_dev(campos))
    P = int(d["means3D"].shape[0])
    M = 0 if d["shs"] is None else int(d["shs"].shape[1])
    N = W * H
    z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device="cuda")
    color, buffer = z(3, H, W), z(NUM_FEATURES, H, W)
    radii, observe = z(max(P, 1), dt=torch.int32), z(max(P, 1), dt=torch.int32)
    h = L.gs2m_ref_create()
    torch.cuda.synchronize()
    R = 0
    if P:  # rasterize_points.cu:84: P == 0 never reaches the rasterizer
        R = L.gs2m_ref_forward(h, P, sh_degree, M, _ptr(d["bg"]), W, H, _ptr(d["means3D"]), _ptr(d["shs"]), _ptr(d["colors_precomp"]),
                               _ptr(d["opacities"]), _ptr(d["scales"]), scale_modifier, _ptr(d["rotations"]), _ptr(d["cov3D_precomp"]),
                               _ptr(d["features"]), _ptr(d["viewmatrix"]), _ptr(d["projmatrix"]), _ptr(d["campos"]), tanfovx, tanfovy,
                               int(prefiltered), feature_count, _ptr(color), _ptr(radii), _ptr(observe), _ptr(buffer))
        if R < 0:
            raise RuntimeError("reference forward failed")
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    out = dict(P=P, W=W, H=H, M=M, num_rendered=R, tiles_x=tiles_x, tiles_y=tiles_y,
               color=color.cpu().numpy(), buffer=buffer.cpu().numpy(), radii=radii[:P].cpu().numpy(), observe=observe[:P].cpu().numpy(),
               _dev=d, _radii_dev=radii, _buffer_dev=buffer,
               _inputs=dict(sh_degree=sh_degree, scale_modifier=scale_modifier, tanfovx=tanfovx, tanfovy=tanfovy, feature_count=feature_count))
    if state and P:
        Pn, Rn = max(P, 1), max(R, 1)
        s = dict(depths=z(Pn), internal_radii=z(Pn, dt=torch.int32), means2D=z(Pn, 2), cov3D=z(Pn, 6), conic_opacity=z(Pn, 4), rgb=z(Pn, 3),
                 tiles_touched=z(Pn, dt=torch.int32), point_offsets=z(Pn, dt=torch.int32), clamped=z(Pn, 3, dt=torch.uint8),
                 keys_sorted=z(Rn, dt=torch.int64), vals_sorted=z(Rn, dt=torch.int32), ranges=z(N, 2, dt=torch.int32),
                 n_contrib=z(N, dt=torch.int32), final_T=z(N))
        order = ("depths", "internal_radii", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched", "point_offsets", "clamped",
                 "keys_sorted", "vals_sorted", "ranges", "n_contrib", "final_T")
        if L.gs2m_ref_state(h, *[_ptr(s[k]) for k in order]) != 0:
            raise RuntimeError("reference state copy failed")
        u32 = lambda t: t.cpu().numpy().view(np.uint32)
        out.update(depths=s["depths"][:P].cpu().numpy(), means2D=s["means2D"][:P].cpu().numpy(), cov3D=s["cov3D"][:P].cpu().numpy(),
                   conic_opacity=s["conic_opacity"][:P].cpu().numpy(), rgb=s["rgb"][:P].cpu().numpy(),
                   clamped=s["clamped"][:P].cpu().numpy(), tiles_touched=u32(s["tiles_touched"])[:P],
                   point_offsets=u32(s["point_offsets"])[:P], keys_sorted=s["keys_sorted"][:R].cpu().numpy().view(np.uint64),
                   vals_sorted=u32(s["vals_sorted"])[:R], ranges=u32(s["ranges"])[:tiles_x * tiles_y],
                   final_T=s["final_T"].cpu().numpy().reshape(H, W), n_contrib=u32(s["n_contrib"]).reshape(H, W))
    return ReferenceForward(h, out)


def backward(fwd, grad_color, grad_buffer):
    """The reference's backward for the forward `fwd`; gradient tensors zero-initialised as rasterize_points.cu:166-176 does
    (the kernels accumulate with atomicAdd)."""
    L = lib()
    d, i = fwd._dev, fwd._inputs
    P, M, W, H = fwd.P, fwd.M, fwd.W, fwd.H
    Pn = max(P, 1)
    z = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device="cuda")
    g = dict(means2D=z(Pn, 4), conics=z(Pn, 2, 2), opacities=z(Pn, 1), colors=z(Pn, 3), means3D=z(Pn, 3), cov3D=z(Pn, 6),
             shs=z(Pn, max(M, 1), 3), scales=z(Pn, 3), rotations=z(Pn, 4), features=z(Pn, NUM_FEATURES))
    gc, gb = _dev(grad_color), _dev(grad_buffer)
    torch.cuda.synchronize()
    if P:
        rc = L.gs2m_ref_backward(fwd._h, P, i["sh_degree"], M, fwd.num_rendered, _ptr(d["bg"]), W, H, _ptr(d["means3D"]), _ptr(d["shs"]),
                                 _ptr(d["colors_precomp"]), _ptr(d["scales"]), i["scale_modifier"], _ptr(d["rotations"]), _ptr(d["cov3D_precomp"]),
                                 _ptr(d["features"]), _ptr(d["viewmatrix"]), _ptr(d["projmatrix"]), _ptr(d["campos"]), i["tanfovx"], i["tanfovy"],
                                 _ptr(fwd._radii_dev), _ptr(fwd._buffer_dev), i["feature_count"], _ptr(gc), _ptr(gb), _ptr(g["means2D"]),
                                 _ptr(g["conics"]), _ptr(g["opacities"]), _ptr(g["colors"]), _ptr(g["means3D"]), _ptr(g["cov3D"]), _ptr(g["shs"]),
                                 _ptr(g["scales"]), _ptr(g["rotations"]), _ptr(g["features"]))
        if rc != 0:
            raise RuntimeError("reference backward failed")
    out = {k: v[:P].cpu().numpy() for k, v in g.items()}
    out["conics"] = out["conics"].reshape(P, 4)
    if M == 0:
        out["shs"] = np.zeros((P, 0, 3), np.float32)
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    m, v, p = _dev(means3D), _dev(viewmatrix), _dev(projmatrix)
    P = int(m.shape[0])
    out = torch.zeros(max(P, 1), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    if P and lib().gs2m_ref_mark_visible(P, _ptr(m), _ptr(v), _ptr(p), _ptr(out)) != 0:
        raise RuntimeError("reference markVisible failed")
    return out[:P].cpu().numpy().astype(bool)


def knn_dist2(points):
    """simple-knn's SimpleKNN::knn (what distCUDA2 returns, spatial.cu:15-26): mean squared distance to the three nearest
    neighbours, (P,) float32."""
    pts = _dev(points)
    P = int(pts.shape[0])
    out = torch.zeros(max(P, 1), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    if P and lib().gs2m_ref_knn(P, _ptr(pts), _ptr(out)) != 0:
        raise RuntimeError("reference knn failed")
    return out[:P].cpu().numpy()


# ---- render-utils' cube-map prefilters (c_src/cubemap.cu): torch CUDA tensors in and out, (6, res, res, C) float32 ----

def _cube(t, ch):
    t = _dev(t)
    assert t.dim() == 4 and t.shape[0] == 6 and t.shape[1] == t.shape[2] and t.shape[3] == ch, tuple(t.shape)
    return t


def _call(name, *args):
    torch.cuda.synchronize()
    if getattr(lib(), "gs2m_ref_" + name)(*args) != 0:
        raise RuntimeError("reference %s failed" % name)


def diffuse_cubemap_fwd(cubemap):
    x = _cube(cubemap, 3)
    out = torch.empty_like(x)
    _call("diffuse_cubemap_fwd", x.shape[1], _ptr(x), _ptr(out))
    return out


def diffuse_cubemap_bwd(cubemap, grad):
    x, g = _cube(cubemap, 3), _cube(grad, 3)
    out = torch.zeros_like(x)
    _call("diffuse_cubemap_bwd", x.shape[1], _ptr(x), _ptr(g), _ptr(out))
    return out


def specular_bounds(res, costheta_cutoff):
    out = torch.zeros(6, res, res, 24, dtype=torch.float32, device="cuda")
    _call("specular_bounds", res, float(costheta_cutoff), _ptr(out))
    return out


def specular_cubemap_fwd(cubemap, bounds, roughness, costheta_cutoff):
    """-> (6, res, res, 4): weighted colour sums and the weight sum (render_utils/ops.py:406 divides them)"""
    x, b = _cube(cubemap, 3), _cube(bounds, 24)
    out = torch.empty(6, x.shape[1], x.shape[1], 4, dtype=torch.float32, device="cuda")
    _call("specular_cubemap_fwd", x.shape[1], _ptr(x), _ptr(b), float(roughness), float(costheta_cutoff), _ptr(out))
    return out


def specular_cubemap_bwd(cubemap, bounds, grad, roughness, costheta_cutoff):
    x, b, g = _cube(cubemap, 3), _cube(bounds, 24), _cube(grad, 4)
    out = torch.zeros_like(x)
    _call("specular_cubemap_bwd", x.shape[1], _ptr(x), _ptr(b), _ptr(g), float(roughness), float(costheta_cutoff), _ptr(out))
    return out


def timed_forward_backward(fwd, grad_color, grad_buffer, n=10):
    """Wall time per forward + backward of the reference build on the inputs of `fwd` (already on the device; the shim
    synchronises the device after each call, the gradient tensors are cleared outside the clock): ms per view."""
    import time
    L = lib()
    d, i = fwd._dev, fwd._inputs
    P, M, W, H = fwd.P, fwd.M, fwd.W, fwd.H
    z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device="cuda")
    color, buffer, radii, observe = z(3, H, W), z(NUM_FEATURES, H, W), z(P, dt=torch.int32), z(P, dt=torch.int32)
    g = [z(P, 4), z(P, 2, 2), z(P, 1), z(P, 3), z(P, 3), z(P, 6), z(P, max(M, 1), 3), z(P, 3), z(P, 4), z(P, NUM_FEATURES)]
    gc, gb = _dev(grad_color), _dev(grad_buffer)
    total = 0.0
    for it in range(n + 2):
        for t in g:
            t.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = L.gs2m_ref_forward(fwd._h, P, i["sh_degree"], M, _ptr(d["bg"]), W, H, _ptr(d["means3D"]), _ptr(d["shs"]), _ptr(d["colors_precomp"]),
                               _ptr(d["opacities"]), _ptr(d["scales"]), i["scale_modifier"], _ptr(d["rotations"]), _ptr(d["cov3D_precomp"]),
                               _ptr(d["features"]), _ptr(d["viewmatrix"]), _ptr(d["projmatrix"]), _ptr(d["campos"]), i["tanfovx"], i["tanfovy"],
                               0, i["feature_count"], _ptr(color), _ptr(radii), _ptr(observe), _ptr(buffer))
        rc = L.gs2m_ref_backward(fwd._h, P, i["sh_degree"], M, R, _ptr(d["bg"]), W, H, _ptr(d["means3D"]), _ptr(d["shs"]),
                                 _ptr(d["colors_precomp"]), _ptr(d["scales"]), i["scale_modifier"], _ptr(d["rotations"]), _ptr(d["cov3D_precomp"]),
                                 _ptr(d["features"]), _ptr(d["viewmatrix"]), _ptr(d["projmatrix"]), _ptr(d["campos"]), i["tanfovx"], i["tanfovy"],
                                 _ptr(radii), _ptr(buffer), i["feature_count"], _ptr(gc), _ptr(gb), *[_ptr(t) for t in g])
        dt = time.perf_counter() - t0
        if R < 0 or rc != 0:
            raise RuntimeError("reference build failed")
        if it >= 2:
            total += dt
    return total / n * 1e3
