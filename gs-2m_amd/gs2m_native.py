"""Loader for the HIP/C-ABI library (csrc/libgs2m_raster.so, include/gs2m_raster.h).

There is NO CPU or PyTorch fallback: if the shared library is missing or cannot be
loaded, every entry point raises.  The library is built in-tree by `build()` (hipcc,
--offload-arch=gfx950) so that it travels with the repository snapshot.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GS2M_LIB", os.path.join(CSRC, "libgs2m_raster.so"))  # GS2M_LIB: A/B builds

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)

EXPORTS = ("gs2m_raster_forward", "gs2m_raster_backward", "gs2m_raster_mark_visible", "gs2m_raster_forward_split_sh", "gs2m_raster_backward_split_sh", "gs2m_knn_dist2",
           "gs2m_debug_layout", "gs2m_debug_tile_sort", "gs2m_raster_forward_token", "gs2m_raster_dense_rows", "gs2m_raster_backward_rows_hint", "gs2m_prealloc_alloc", "gs2m_set_debug", "gs2m_set_markers", "gs2m_stage_name", "gs2m_set_reference_binning", "gs2m_set_spin_wait", "gs2m_set_sort_tickets", "gs2m_set_tile_sort_policy", "gs2m_pack_features_forward", "gs2m_pack_features_backward", "gs2m_gbuffer_post_forward",
           "gs2m_gbuffer_post_backward", "gs2m_gbuffer_maps_backward", "gs2m_sobel_normal_forward", "gs2m_sobel_normal_backward", "gs2m_activate_forward", "gs2m_activate_backward", "gs2m_texture_cube_forward", "gs2m_texture_cube_backward", "gs2m_texture_2d_clamp_forward", "gs2m_texture_2d_clamp_backward", "gs2m_diffuse_cubemap_forward", "gs2m_diffuse_cubemap_backward", "gs2m_cubemap_texel_table", "gs2m_specular_cubemap_forward", "gs2m_specular_cubemap_backward", "gs2m_specular_cubemap_normalized_forward", "gs2m_specular_cubemap_normalized_backward", "gs2m_pbr_shade_forward", "gs2m_pbr_shade_backward", "gs2m_patch_ncc_forward", "gs2m_patch_ncc_backward", "gs2m_patch_ncc_roughness", "gs2m_grid_sample_border_forward", "gs2m_grid_sample_border_backward", "gs2m_mv_geo_forward", "gs2m_mv_geo_backward", "gs2m_mvs_set_deterministic", "gs2m_mvs_get_deterministic", "gs2m_adam_step", "gs2m_ssim_forward", "gs2m_ssim_backward", "gs2m_profile_mode", "gs2m_profile_sampling", "gs2m_profile_collect", "gs2m_version",
           # include/gs2m_loss.h (round 3: the loss tail of the training iteration)
           "gs2m_affine_mean", "gs2m_densification_stats", "gs2m_edge_gradient", "gs2m_image_loss_backward", "gs2m_image_loss_forward", "gs2m_loss_workspace_bytes", "gs2m_mv_geo_loss_backward", "gs2m_mv_geo_loss_forward", "gs2m_mv_take_backward", "gs2m_mv_take_forward", "gs2m_ncc_tail_backward", "gs2m_ncc_tail_forward", "gs2m_subset_thin", "gs2m_subset_remove", "gs2m_pbr_inputs_backward", "gs2m_pbr_inputs_forward", "gs2m_plane_loss_backward", "gs2m_plane_loss_forward", "gs2m_ssim_backward_uniform", "gs2m_tv_loss_backward", "gs2m_tv_loss_forward")

STAGES = ("preprocess", "unused1", "scan", "emit", "tile_sort", "lists", "blend_fwd", "unused7", "blend_bwd",
          "gaussian_bwd")

_lib = None


class Prealloc(C.Structure):
    """include/gs2m_raster.h: gs2m_prealloc (user block of gs2m_prealloc_alloc)"""
    _fields_ = [("ptr", C.c_void_p), ("capacity", C.c_size_t), ("fallback", ALLOC_FN), ("fallback_user", C.c_void_p),
                ("used_fallback", C.c_int), ("requested", C.c_size_t)]


class Layout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "geom_bytes", "rec", "tiles_touched", "depth_key", "rect", "gauss_rows", "clamped", "wave_rowbase", "counters",
        "binning_bytes", "point_list", "tile_keys", "qlist", "qrow",
        "image_bytes", "final_T", "n_contrib", "ranges", "qcount")]


def build(jobs=8, force=False):
    """Compile every HIP translation unit for gfx950 and link libgs2m_raster.so."""
    cmd = ["make", "-C", CSRC, "-j", str(jobs)]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"gs2m: HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (needs hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    p, i, f = C.c_void_p, C.c_int, C.c_float
    L.gs2m_raster_forward.restype = i
    L.gs2m_raster_forward.argtypes = [ALLOC_FN, p, ALLOC_FN, p, ALLOC_FN, p, i, i, i, p, i, i, p, p, p, p, p, f, p, p,
                                      p, p, p, p, f, f, i, i, p, p, p, p, p]
    L.gs2m_raster_backward.restype = i
    L.gs2m_raster_backward.argtypes = [i, i, i, i, p, i, i, p, p, p, p, f, p, p, p, p, p, p, f, f, p, p, p, p, p, i,
                                       p, p, p, p, p, p, p, p, p, p, p, p, ALLOC_FN, p, p]
    L.gs2m_raster_forward_split_sh.restype = i
    L.gs2m_raster_forward_split_sh.argtypes = [ALLOC_FN, p, ALLOC_FN, p, ALLOC_FN, p, i, i, i, p, i, i, p, p, p, p, p, p, f, p, p,
                                               p, p, p, p, f, f, i, i, p, p, p, p, p]
    L.gs2m_raster_backward_split_sh.restype = i
    L.gs2m_raster_backward_split_sh.argtypes = [i, i, i, i, p, i, i, p, p, p, p, p, f, p, p, p, p, p, p, f, f, p, p, p, p, p, i,
                                                p, p, p, p, p, p, p, p, p, p, p, p, p, ALLOC_FN, p, p]
    L.gs2m_raster_mark_visible.restype = i
    L.gs2m_raster_mark_visible.argtypes = [i, p, p, p, p, p]
    L.gs2m_knn_dist2.restype = i
    L.gs2m_knn_dist2.argtypes = [i, p, p, ALLOC_FN, p, p]
    L.gs2m_debug_layout.restype = i
    L.gs2m_debug_layout.argtypes = [i, i, i, i, C.POINTER(Layout)]
    L.gs2m_raster_forward_token.restype = C.c_ulonglong
    L.gs2m_raster_forward_token.argtypes = []
    L.gs2m_raster_dense_rows.restype = C.c_longlong
    L.gs2m_raster_dense_rows.argtypes = [C.c_ulonglong]
    L.gs2m_raster_backward_rows_hint.restype = i
    L.gs2m_raster_backward_rows_hint.argtypes = [C.c_longlong]
    L.gs2m_debug_tile_sort.restype = i
    L.gs2m_debug_tile_sort.argtypes = [i] + [p] * 11
    L.gs2m_set_debug.restype = i
    L.gs2m_set_debug.argtypes = [i]
    L.gs2m_set_markers.restype = i
    L.gs2m_set_markers.argtypes = [i]
    L.gs2m_stage_name.restype = C.c_char_p
    L.gs2m_stage_name.argtypes = [i]
    L.gs2m_set_reference_binning.restype = i
    L.gs2m_set_reference_binning.argtypes = [i]
    L.gs2m_set_spin_wait.restype = i
    L.gs2m_set_spin_wait.argtypes = [i]
    L.gs2m_set_sort_tickets.restype = i
    L.gs2m_set_sort_tickets.argtypes = [i]
    L.gs2m_set_tile_sort_policy.restype = i
    L.gs2m_set_tile_sort_policy.argtypes = [i]
    L.gs2m_pack_features_forward.restype = i
    L.gs2m_pack_features_forward.argtypes = [i, p, p, p, p, p, p, p, p, i, i, p, p]
    L.gs2m_pack_features_backward.restype = i
    L.gs2m_pack_features_backward.argtypes = [i, p, p, p, p, p, i, i, p, p, p, p, p, p, p]
    L.gs2m_gbuffer_post_forward.restype = i
    L.gs2m_gbuffer_post_forward.argtypes = [i, i, p, p, p, i, p, p, p, p]
    L.gs2m_gbuffer_post_backward.restype = i
    L.gs2m_gbuffer_post_backward.argtypes = [i, i, p, p, p, i, p, p, p, p]
    L.gs2m_gbuffer_maps_backward.restype = i
    L.gs2m_gbuffer_maps_backward.argtypes = [i, i, p, p, p, i, p, p, p, p, p, p, p, p, p, p]
    L.gs2m_sobel_normal_forward.restype = i
    L.gs2m_sobel_normal_forward.argtypes = [i, i, p, p, p, p, f, f, f, f, p, p]
    L.gs2m_sobel_normal_backward.restype = i
    L.gs2m_sobel_normal_backward.argtypes = [i, i, p, p, p, p, f, f, f, f, p, p, p, p]
    L.gs2m_activate_forward.restype = i
    L.gs2m_activate_forward.argtypes = [i] + [p] * 13
    L.gs2m_activate_backward.restype = i
    L.gs2m_activate_backward.argtypes = [i] + [p] * 19
    L.gs2m_texture_cube_forward.restype = i
    L.gs2m_texture_cube_forward.argtypes = [i, i, i, p, p, p, p, p, p]
    L.gs2m_texture_cube_backward.restype = i
    L.gs2m_texture_cube_backward.argtypes = [i, i, i, p, p, p, p, p, i, p]
    L.gs2m_texture_2d_clamp_forward.restype = i
    L.gs2m_texture_2d_clamp_forward.argtypes = [i, i, i, i, p, p, p, p]
    L.gs2m_texture_2d_clamp_backward.restype = i
    L.gs2m_texture_2d_clamp_backward.argtypes = [i, i, i, i, p, p, p, p]
    L.gs2m_diffuse_cubemap_forward.restype = i
    L.gs2m_diffuse_cubemap_forward.argtypes = [i, p, p, p]
    L.gs2m_diffuse_cubemap_backward.restype = i
    L.gs2m_diffuse_cubemap_backward.argtypes = [i, p, p, p]
    L.gs2m_specular_cubemap_forward.restype = i
    L.gs2m_cubemap_texel_table.restype = i
    L.gs2m_cubemap_texel_table.argtypes = [i, p, p]
    L.gs2m_specular_cubemap_forward.argtypes = [i, f, f, p, p, p, p]
    L.gs2m_specular_cubemap_backward.restype = i
    L.gs2m_specular_cubemap_backward.argtypes = [i, f, f, p, p, p, p]
    L.gs2m_pbr_shade_forward.restype = i
    L.gs2m_pbr_shade_forward.argtypes = [i, p, p, p, p, p, p, i, i, p, i, i, p, p, f, f, p, p, p, p, p]
    L.gs2m_pbr_shade_backward.restype = i
    L.gs2m_pbr_shade_backward.argtypes = [i, p, p, p, p, p, p, i, i, p, i, i, p, p, f, f, p, p, p, p, p, i, p]
    L.gs2m_patch_ncc_forward.restype = i
    L.gs2m_patch_ncc_forward.argtypes = [i, p, p, p, p, p, i, i, p, p, p, f, i, p, p]
    L.gs2m_patch_ncc_backward.restype = i
    L.gs2m_patch_ncc_backward.argtypes = [i, p, p, p, p, p, i, i, p, p, p, f, i, p, p, p, p]
    L.gs2m_patch_ncc_roughness.restype = i
    L.gs2m_patch_ncc_roughness.argtypes = [i, p, p, p, p, p, i, i, p, p, p, f, i, p, p, p, p]
    L.gs2m_specular_cubemap_normalized_forward.restype = i
    L.gs2m_specular_cubemap_normalized_forward.argtypes = [i, f, f, p, p, p, p, p]
    L.gs2m_specular_cubemap_normalized_backward.restype = i
    L.gs2m_specular_cubemap_normalized_backward.argtypes = [i, f, f, p, p, p, p, p, p]
    L.gs2m_grid_sample_border_forward.restype = i
    L.gs2m_grid_sample_border_forward.argtypes = [i, i, i, i, p, p, p, p]
    L.gs2m_mvs_set_deterministic.restype = None
    L.gs2m_mvs_set_deterministic.argtypes = [i]
    L.gs2m_mvs_get_deterministic.restype = i
    L.gs2m_mvs_get_deterministic.argtypes = []
    L.gs2m_grid_sample_border_backward.restype = i
    L.gs2m_grid_sample_border_backward.argtypes = [i, i, i, i, p, p, p, p, p, p]
    L.gs2m_mv_geo_forward.restype = i
    L.gs2m_mv_geo_forward.argtypes = [i, i, i, i, p, p, p, p, p, p, p, p, p, p, f, p, p, p, p]
    L.gs2m_mv_geo_backward.restype = i
    L.gs2m_mv_geo_backward.argtypes = [i, i, i, i, p, p, p, p, p, p, p, p, p, p, f, p, p, p, p, p, p, p]
    L.gs2m_adam_step.restype = i
    L.gs2m_adam_step.argtypes = [i, p, C.c_double, C.c_double, C.c_double, p]
    L.gs2m_ssim_forward.restype = i
    L.gs2m_ssim_forward.argtypes = [i, i, i, i, f, f, p, p, p, p, p, p, p]
    L.gs2m_ssim_backward.restype = i
    L.gs2m_ssim_backward.argtypes = [i, i, i, i, p, p, p, p, p, p, p, p]
    L.gs2m_profile_mode.restype = i
    L.gs2m_loss_workspace_bytes.argtypes = []
    L.gs2m_loss_workspace_bytes.restype = i
    L.gs2m_edge_gradient.argtypes = [i, i, p, p, p, p, p]
    L.gs2m_edge_gradient.restype = i
    L.gs2m_image_loss_forward.argtypes = [i, i, p, i, p, p, p, p, p, p, p, p, f, f, p, p, p, p]
    L.gs2m_image_loss_forward.restype = i
    L.gs2m_image_loss_backward.argtypes = [i, i, p, i, p, p, p, p, p, p, p, f, f, p, p, p, p, p, p]
    L.gs2m_image_loss_backward.restype = i
    L.gs2m_pbr_inputs_forward.argtypes = [i, i, p, p, p, p, p, f, f, p, p, p, p, p]
    L.gs2m_pbr_inputs_forward.restype = i
    L.gs2m_pbr_inputs_backward.argtypes = [i, i, p, p, p, p]
    L.gs2m_pbr_inputs_backward.restype = i
    L.gs2m_tv_loss_forward.argtypes = [i, i, i, p, p, p, i, f, p, p, p]
    L.gs2m_tv_loss_forward.restype = i
    L.gs2m_tv_loss_backward.argtypes = [i, i, i, p, p, p, i, f, p, p, p]
    L.gs2m_tv_loss_backward.restype = i
    L.gs2m_mv_geo_loss_forward.argtypes = [i, p, p, p, f, f, f, f, p, p, p, p, p]
    L.gs2m_mv_geo_loss_forward.restype = i
    L.gs2m_mv_geo_loss_backward.argtypes = [i, p, p, p, f, f, f, f, p, p, p, p, p]
    L.gs2m_mv_geo_loss_backward.restype = i
    L.gs2m_mv_take_forward.argtypes = [i, p, i, i, p, p, p, p, p, p, p, p]
    L.gs2m_mv_take_forward.restype = i
    L.gs2m_mv_take_backward.argtypes = [i, p, i, i, p, p, p, p, p]
    L.gs2m_mv_take_backward.restype = i
    L.gs2m_ncc_tail_forward.argtypes = [i, p, p, p, p, p]
    L.gs2m_ncc_tail_forward.restype = i
    L.gs2m_ncc_tail_backward.argtypes = [i, p, p, p, p, p, p]
    L.gs2m_ncc_tail_backward.restype = i
    L.gs2m_subset_thin.argtypes = [i, p, i, C.c_ulonglong, p, i, p, p, p]
    L.gs2m_subset_thin.restype = i
    L.gs2m_subset_remove.argtypes = [i, i, C.c_ulonglong, p, p, p]
    L.gs2m_subset_remove.restype = i
    L.gs2m_affine_mean.argtypes = [C.c_longlong, p, f, f, p, p, p]
    L.gs2m_affine_mean.restype = i
    L.gs2m_ssim_backward_uniform.argtypes = [i, i, i, i, p, p, p, f, f, p, p, p, p, p]
    L.gs2m_ssim_backward_uniform.restype = i
    L.gs2m_plane_loss_forward.argtypes = [i, p, i, p, f, p, p, p]
    L.gs2m_plane_loss_forward.restype = i
    L.gs2m_plane_loss_backward.argtypes = [i, p, i, p, f, p, p, p, p]
    L.gs2m_plane_loss_backward.restype = i
    L.gs2m_densification_stats.argtypes = [i, p, p, p, p, p, p, p, p, p]
    L.gs2m_densification_stats.restype = i
    L.gs2m_profile_mode.argtypes = [i]
    L.gs2m_profile_sampling.restype = i
    L.gs2m_profile_sampling.argtypes = [i]
    L.gs2m_profile_collect.restype = i
    L.gs2m_profile_collect.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int), i]
    L.gs2m_version.restype = C.c_char_p
    _lib = L
    return L


ERRORS = {-1: "invalid argument", -2: "HIP runtime error", -3: "scratch allocation failed", -4: "unsupported size",
          -5: "Point culled! This point should have been prefiltered (prefiltered=True and a Gaussian has view z <= 0.2; the reference traps the device here, cuda_rasterizer/auxiliary.h:155-158)"}


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """`with torch.cuda.device(device)` without its cost when `device` is already the current one (every call of every
    wrapper: ~10 us of host time each, 0.2 ms per training iteration)."""
    import torch
    idx = device.index if hasattr(device, "index") else device
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def stream_ptr(device=None):
    """The raw hipStream_t of torch's current stream on `device` (the current device when None) as an int -- one C call
    instead of building a torch.cuda.Stream object per launch."""
    import torch
    idx = None if device is None else (device.index if hasattr(device, "index") else device)
    if idx is None:
        idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def check(rc, what):
    if rc <= -100:  # debug mode: GS2M_ERR_STAGE(stage)
        raise RuntimeError(f"gs2m: {what} failed in stage `{lib().gs2m_stage_name(-100 - rc).decode()}` (debug mode)")
    if rc < 0:
        raise RuntimeError(f"gs2m: {what} failed: {ERRORS.get(rc, rc)}")
    return rc


def set_debug(on):
    """1: synchronize and check the stream after every pipeline stage; a fault is raised naming the stage."""
    check(lib().gs2m_set_debug(int(bool(on))), "gs2m_set_debug")


def set_markers(on):
    """1: roctx ranges around the pipeline stages (rocprofv3 --marker-trace)."""
    check(lib().gs2m_set_markers(int(bool(on))), "gs2m_set_markers")


def reset_modes():
    """Every process-wide switch back to its default (tests/conftest.py calls this after each test)."""
    L = lib()
    L.gs2m_set_reference_binning(0); L.gs2m_set_spin_wait(1); L.gs2m_set_debug(0); L.gs2m_set_sort_tickets(0); L.gs2m_set_tile_sort_policy(0)
    L.gs2m_set_markers(0); L.gs2m_profile_mode(0)


def debug_layout(P, R, W, H):
    lay = Layout()
    check(lib().gs2m_debug_layout(P, R, W, H, C.byref(lay)), "gs2m_debug_layout")
    return lay


def set_sort_tickets(on):
    """True: the tile sort takes its tile ids from an atomic ticket (other kernels -- RCCL's -- may be resident on the device)."""
    check(lib().gs2m_set_sort_tickets(int(bool(on))), "gs2m_set_sort_tickets")


def set_tile_sort_policy(policy):
    """0: by tile count (default); 1: a workgroup per tile; 2: a wave per tile (spans of up to 1024 entries); 3: a wave per tile, the
    spans of 513 .. 1024 entries left to the workgroup kernel that takes the longer ones (0 and 2 choose that by themselves in a frame
    whose average span is at most 400 entries)."""
    check(lib().gs2m_set_tile_sort_policy(int(policy)), "gs2m_set_tile_sort_policy")


def set_spin_wait(on):
    """Forward: poll the pinned num_rendered (default) or sleep in hipStreamSynchronize."""
    check(lib().gs2m_set_spin_wait(int(bool(on))), "gs2m_set_spin_wait")


def profile_mode(mode, every=1):
    """0 off, 1 blend kernels only, 2 every stage, 3 backward blend only (HIP events on the launch stream); `every`: mode 3
    brackets every `every`-th launch (an event pair leaves ~12 us of bubble around the kernel)."""
    check(lib().gs2m_profile_sampling(int(every)), "gs2m_profile_sampling")
    check(lib().gs2m_profile_mode(int(mode)), "gs2m_profile_mode")


def profile_collect():
    """-> {stage: (total_ms, launches)} since the last collect."""
    n = len(STAGES)
    ms = (C.c_float * n)()
    cnt = (C.c_int * n)()
    check(lib().gs2m_profile_collect(ms, cnt, n), "gs2m_profile_collect")
    return {STAGES[k]: (float(ms[k]), int(cnt[k])) for k in range(n)}


def set_reference_binning(on):
    """True: emit exactly the reference's tile rectangles (bit-identical sorted lists, for the parity tests
    of the integer artefacts); False (default): drop tiles the alpha >= 1/255 ellipse cannot reach."""
    check(lib().gs2m_set_reference_binning(1 if on else 0), "gs2m_set_reference_binning")
