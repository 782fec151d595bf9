"""ctypes front-end of the CPU oracle (oracle/gs2m_oracle.c).

TEST INFRASTRUCTURE ONLY -- pinned to the reference build oracle/_ref by tests/test_reference_gpu.py (see the C file's header).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (gs-2m_amd/) never does.

All arrays are numpy, C-contiguous; floats are fp32, indices int32/uint32/uint64.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs2m_oracle.so")
_lib = None

NUM_FEATURES = 10


class _State(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("W", C.c_int), ("H", C.c_int), ("R", C.c_int),
        ("tiles_x", C.c_int), ("tiles_y", C.c_int), ("sort_bits", C.c_int), ("pad_", C.c_int),
        ("depths", C.POINTER(C.c_float)), ("means2D", C.POINTER(C.c_float)),
        ("cov3D", C.POINTER(C.c_float)), ("conic_opacity", C.POINTER(C.c_float)),
        ("rgb", C.POINTER(C.c_float)), ("clamped", C.POINTER(C.c_uint8)),
        ("radii", C.POINTER(C.c_int)), ("tiles_touched", C.POINTER(C.c_uint32)),
        ("point_offsets", C.POINTER(C.c_uint32)),
        ("keys_unsorted", C.POINTER(C.c_uint64)), ("vals_unsorted", C.POINTER(C.c_uint32)),
        ("keys_sorted", C.POINTER(C.c_uint64)), ("vals_sorted", C.POINTER(C.c_uint32)),
        ("ranges", C.POINTER(C.c_uint32)), ("final_T", C.POINTER(C.c_float)),
        ("n_contrib", C.POINTER(C.c_uint32)),
    ]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "gs2m_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgs2m_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def use_native_build():
    """bench.py's cpu_baseline leg only: compile the same source at -O3 -march=native ON THIS MACHINE
    (oracle/Makefile: libgs2m_oracle_native.so) and route this module's calls to it.  Returns the flags used, or None
    when the build is not possible (the -O2 checker build is then timed instead).  Never used by the parity tests: the
    checker stays the portable -O2 build."""
    global _lib
    path = os.path.join(_HERE, "libgs2m_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgs2m_oracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = None
        lib(path)
        return "-O3 -march=native -fopenmp"
    except Exception:
        _lib = None
        return None


def lib(path=None):
    global _lib
    if _lib is None:
        if path is None and not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(path or _LIB_PATH)
        _lib.gs2m_oracle_forward.restype = C.POINTER(_State)
        _lib.gs2m_oracle_free.argtypes = [C.POINTER(_State)]
        _lib.gs2m_oracle_higher_msb.restype = C.c_uint32
        _lib.gs2m_oracle_morton.restype = C.c_uint32
    return _lib


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleForward:
    """Result of one oracle forward: API outputs plus every internal artefact."""

    def __init__(self, handle, out):
        self._h = handle
        self.__dict__.update(out)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib is not None:
                _lib.gs2m_oracle_free(self._h)
        except Exception:  # interpreter shutdown
            pass
        self._h = None


def _copy(ptr, n, dtype):
    if n == 0:
        return np.zeros((0,), dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def forward(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, features=None, bg, viewmatrix, projmatrix, campos, W, H,
            tanfovx, tanfovy, sh_degree=0, scale_modifier=1.0, prefiltered=False, feature_count=0):
    L = lib()
    means3D = _f32(means3D); opacities = _f32(opacities)
    shs = _f32(shs); colors_precomp = _f32(colors_precomp); scales = _f32(scales)
    rotations = _f32(rotations); cov3D_precomp = _f32(cov3D_precomp); features = _f32(features)
    bg = _f32(bg); viewmatrix = _f32(viewmatrix); projmatrix = _f32(projmatrix); campos = _f32(campos)
    P = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    N = W * H
    color = np.zeros((3, H, W), np.float32)
    buffer = np.zeros((NUM_FEATURES, H, W), np.float32)
    radii = np.zeros((max(P, 1),), np.int32)
    observe = np.zeros((max(P, 1),), np.int32)
    h = L.gs2m_oracle_forward(
        C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(bg), C.c_int(W), C.c_int(H),
        _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales), C.c_float(scale_modifier),
        _p(rotations), _p(cov3D_precomp), _p(features), _p(viewmatrix), _p(projmatrix), _p(campos),
        C.c_float(tanfovx), C.c_float(tanfovy), C.c_int(int(prefiltered)), C.c_int(feature_count),
        _p(color), _p(radii), _p(observe), _p(buffer))
    s = h.contents
    R = s.R
    Tn = s.tiles_x * s.tiles_y
    out = dict(
        P=P, W=W, H=H, M=M, num_rendered=R, tiles_x=s.tiles_x, tiles_y=s.tiles_y, sort_bits=s.sort_bits,
        color=color, buffer=buffer, radii=radii[:P], observe=observe[:P],
        depths=_copy(s.depths, P, np.float32),
        means2D=_copy(s.means2D, 2 * P, np.float32).reshape(P, 2),
        cov3D=_copy(s.cov3D, 6 * P, np.float32).reshape(P, 6),
        conic_opacity=_copy(s.conic_opacity, 4 * P, np.float32).reshape(P, 4),
        rgb=_copy(s.rgb, 3 * P, np.float32).reshape(P, 3),
        clamped=_copy(s.clamped, 3 * P, np.uint8).reshape(P, 3),
        tiles_touched=_copy(s.tiles_touched, P, np.uint32),
        point_offsets=_copy(s.point_offsets, P, np.uint32),
        keys_unsorted=_copy(s.keys_unsorted, R, np.uint64),
        vals_unsorted=_copy(s.vals_unsorted, R, np.uint32),
        keys_sorted=_copy(s.keys_sorted, R, np.uint64),
        vals_sorted=_copy(s.vals_sorted, R, np.uint32),
        ranges=_copy(s.ranges, 2 * Tn, np.uint32).reshape(Tn, 2),
        final_T=_copy(s.final_T, N, np.float32).reshape(H, W),
        n_contrib=_copy(s.n_contrib, N, np.uint32).reshape(H, W),
        _inputs=dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales,
                     rotations=rotations, cov3D_precomp=cov3D_precomp, features=features, bg=bg,
                     viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos, tanfovx=tanfovx,
                     tanfovy=tanfovy, sh_degree=sh_degree, scale_modifier=scale_modifier,
                     feature_count=feature_count),
    )
    return OracleForward(h, out)


def backward(fwd, grad_color, grad_buffer):
    """Gradients for the forward `fwd` (an OracleForward) given dL/dcolor (3,H,W), dL/dbuffer (10,H,W)."""
    L = lib()
    i = fwd._inputs
    P, M, W, H = fwd.P, fwd.M, fwd.W, fwd.H
    gc = _f32(grad_color); gb = _f32(grad_buffer)
    Pn = max(P, 1)
    z = lambda *sh: np.zeros(sh, np.float32)
    g = dict(means2D=z(Pn, 4), conics=z(Pn, 4), opacities=z(Pn, 1), colors=z(Pn, 3), means3D=z(Pn, 3),
             cov3D=z(Pn, 6), shs=z(Pn, max(M, 1), 3), scales=z(Pn, 3), rotations=z(Pn, 4),
             features=z(Pn, NUM_FEATURES))
    radii = np.ascontiguousarray(fwd.radii if P else np.zeros(1, np.int32), dtype=np.int32)
    L.gs2m_oracle_backward(
        fwd._h, C.c_int(P), C.c_int(i["sh_degree"]), C.c_int(M), _p(i["bg"]), C.c_int(W), C.c_int(H),
        _p(i["means3D"]), _p(i["shs"]), _p(i["colors_precomp"]), _p(i["scales"]), C.c_float(i["scale_modifier"]),
        _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["features"]), _p(i["viewmatrix"]), _p(i["projmatrix"]),
        _p(i["campos"]), C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]), _p(radii), C.c_int(i["feature_count"]),
        _p(gc), _p(gb), _p(g["means2D"]), _p(g["conics"]), _p(g["opacities"]), _p(g["colors"]), _p(g["means3D"]),
        _p(g["cov3D"]), _p(g["shs"]), _p(g["scales"]), _p(g["rotations"]), _p(g["features"]))
    out = {k: v[:P] for k, v in g.items()}
    if M == 0:
        out["shs"] = np.zeros((P, 0, 3), np.float32)
    return out


def backward_pergaussian(fwd, dL_dmeans2D, dL_dconics, dL_dcolors):
    """The per-Gaussian half of the backward (cov2D / projection / SH / cov3D chains) applied to GIVEN per-Gaussian
    sums (means2D (P,4), conics (P,4) = (P,2,2) flattened, colours (P,3)) -- e.g. another implementation's."""
    L = lib()
    i = fwd._inputs
    P, M, W, H = fwd.P, fwd.M, fwd.W, fwd.H
    Pn = max(P, 1)
    z = lambda *sh: np.zeros(sh, np.float32)
    g = dict(means3D=z(Pn, 3), cov3D=z(Pn, 6), shs=z(Pn, max(M, 1), 3), scales=z(Pn, 3), rotations=z(Pn, 4))
    radii = np.ascontiguousarray(fwd.radii if P else np.zeros(1, np.int32), dtype=np.int32)
    m2 = _f32(np.asarray(dL_dmeans2D).reshape(-1, 4)); co = _f32(np.asarray(dL_dconics).reshape(-1, 4))
    cl = _f32(np.asarray(dL_dcolors).reshape(-1, 3))
    L.gs2m_oracle_backward_pergaussian(
        fwd._h, C.c_int(P), C.c_int(i["sh_degree"]), C.c_int(M), C.c_int(W), C.c_int(H), _p(i["means3D"]), _p(i["shs"]),
        _p(i["scales"]), C.c_float(i["scale_modifier"]), _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["viewmatrix"]),
        _p(i["projmatrix"]), _p(i["campos"]), C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]), _p(radii), _p(m2), _p(co),
        _p(cl), _p(g["means3D"]), _p(g["cov3D"]), _p(g["shs"]), _p(g["scales"]), _p(g["rotations"]))
    out = {k: v[:P] for k, v in g.items()}
    if M == 0:
        out["shs"] = np.zeros((P, 0, 3), np.float32)
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros((max(P, 1),), np.uint8)
    lib().gs2m_oracle_mark_visible(C.c_int(P), _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(out))
    return out[:P].astype(bool)


def knn_dist2(points):
    points = _f32(points)
    P = points.shape[0]
    out = np.zeros((max(P, 1),), np.float32)
    lib().gs2m_oracle_knn_dist2(C.c_int(P), _p(points), _p(out))
    return out[:P]


def morton(coord, minn, maxx):
    return int(lib().gs2m_oracle_morton(_p(_f32(coord)), _p(_f32(minn)), _p(_f32(maxx))))


def higher_msb(n):
    return int(lib().gs2m_oracle_higher_msb(C.c_uint32(n)))
