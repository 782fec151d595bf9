"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares.
No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"^\s*(?:const\s+)?(?:unsigned\s+long\s+long|long\s+long|int|char\s*\*|void)\s*\*?\s*(gs2m_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for n in ("gs2m_raster_forward", "gs2m_raster_backward", "gs2m_raster_mark_visible", "gs2m_knn_dist2"):
        assert n in names


def test_library_exports_every_declared_symbol():
    import gs2m_native
    if not os.path.exists(gs2m_native.LIB_PATH):
        gs2m_native.build()
    lib = ctypes.CDLL(gs2m_native.LIB_PATH)
    for n in _declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    # the list build() checks is the whole declared surface: a header added without its EXPORTS entry fails here
    assert set(gs2m_native.EXPORTS) == set(_declared_functions())
    assert b"gfx950" in ctypes.cast(lib.gs2m_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_reference_build_recipe_and_library():
    """oracle/ref_build/: the recipe and this repository's shim are committed, no translated reference text is; where the
    library has been built (the build container; it travels prebuilt), it loads without a GPU and exports the shim's
    entry points"""
    d = os.path.join(ROOT, "oracle", "ref_build")
    assert sorted(os.listdir(d)) == ["Makefile", "ref_shim.hip", "ref_shim_cubemap.hip", "ref_shim_ssim.hip"]
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if os.path.isdir(ref_dir):
        assert all(f.endswith(".so") for f in os.listdir(ref_dir)), "only built libraries belong in oracle/_ref"
    so = os.path.join(ref_dir, "libgs2m_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libgs2m_ref.so not built here")
    lib = ctypes.CDLL(so)
    for n in ("gs2m_ref_create", "gs2m_ref_destroy", "gs2m_ref_forward", "gs2m_ref_backward", "gs2m_ref_state", "gs2m_ref_mark_visible", "gs2m_ref_knn",
              "gs2m_ref_diffuse_cubemap_fwd", "gs2m_ref_diffuse_cubemap_bwd", "gs2m_ref_specular_bounds", "gs2m_ref_specular_cubemap_fwd",
              "gs2m_ref_specular_cubemap_bwd"):
        assert hasattr(lib, n), n


def test_product_package_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under gs-2m_amd/ may import, include or link it."""
    pkg = os.path.join(ROOT, "gs-2m_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "gs2m_oracle" not in txt and "oracle/" not in txt, f
    # nor do the scripts under tools/ (reports that use the checkers live in tests/)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_op_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import helpers as Hh
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = Hh.make_scene(8, 32, 32)
    r = GaussianRasterizer(Hh.settings_for(sc, "cpu"))
    g = sc["g"]
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(g["means3D"], torch.zeros(8, 4), g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
          features=g["features"])


def test_render_ops_refuse_cpu_tensors():
    """the fused render() pre/post-processing has no CPU path either: CPU tensors raise instead of silently computing"""
    import torch
    import gs2m_render_ops as R
    P = 4
    with pytest.raises(RuntimeError, match="no CPU path"):
        R.pack_features(torch.zeros(P, 3), torch.ones(P, 3), torch.ones(P, 4), torch.zeros(P, 3), torch.zeros(P, 1),
                        torch.zeros(P, 1), torch.zeros(3), torch.eye(4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        R.gbuffer_post(torch.zeros(10, 4, 4), torch.zeros(16, 3), torch.eye(4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        R.sobel_normal(torch.ones(4, 4), torch.ones(4, 4), torch.zeros(3), torch.eye(4), 1.0, 1.0, 2.0, 2.0)


def test_fused_adam_refuses_cpu_tensors_and_unsupported_modes():
    import torch
    import gs2m_optim
    p = torch.nn.Parameter(torch.zeros(4))
    opt = gs2m_optim.Adam([p], lr=0.1)
    assert set(opt.param_groups[0].keys()) == set(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]).param_groups[0].keys())
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        opt.step()
    with pytest.raises(NotImplementedError):
        gs2m_optim.Adam([p], weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        gs2m_optim.Adam([p], amsgrad=True)


def test_fused_ssim_refuses_cpu_tensors():
    import torch
    from fused_ssim import fused_ssim
    with pytest.raises(RuntimeError, match="no CPU path"):
        fused_ssim(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16))


def test_every_header_is_valid_c99():
    """The boundary is a C ABI: each header must compile as plain C on its own."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if fn.endswith(".h"):
            r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", "-"],
                               input=f'#include "{os.path.join(inc, fn)}"\nint main(void) {{ return 0; }}\n', text=True, capture_output=True)
            assert r.returncode == 0, fn + "\n" + r.stderr
