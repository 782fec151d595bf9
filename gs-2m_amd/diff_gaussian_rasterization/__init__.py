"""MI355X-native drop-in for the reference's `diff_gaussian_rasterization` package.

Same public surface as /root/reference/submodules/diff-gaussian-rasterization/
diff_gaussian_rasterization/__init__.py:
  GaussianRasterizationSettings (NamedTuple, 12 fields, :143-155)
  GaussianRasterizer(nn.Module).forward(...) -> (color, radii, observe, buffer) and
  .markVisible(positions) (:157-218)
  rasterize_gaussians / _RasterizeGaussians (autograd.Function, 10 inputs, 9 grads + None,
  :17-141)
and a `_C` namespace with the three functions of the reference's pybind module
(ext.cpp:15-19; rasterize_points.cu:30-219), implemented on top of the C ABI in
include/gs2m_raster.h through ctypes.  PyTorch only provides device memory and the stream.
There is no fallback: tensors must live on a HIP device and the HIP library must be built.
"""
from typing import NamedTuple

import ctypes as C
import threading

import os

import torch
import torch.nn as nn

import gs2m_arena as _arena
import gs2m_native as _native

NUM_CHANNELS = 3
NUM_FEATURES = 10


def _ptr(t):
    """Device pointer or NULL for the reference's 'missing optional' encoding (an empty tensor)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _f32c(t, name):
    if t is None or t.numel() == 0:
        return t
    if not t.is_cuda:
        raise RuntimeError(f"gs2m rasterizer: `{name}` must be on a HIP (cuda) device; there is no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"gs2m rasterizer: `{name}` must be float32, got {t.dtype}")
    return t.contiguous()


class _Alloc:
    """Scratch allocator handed to the C ABI (mirrors resizeFunctional, rasterize_points.cu:22-28)."""

    def __init__(self, device):
        self.device = device
        self.tensor = None
        self.cb = _native.ALLOC_FN(self._alloc)  # creating a ctypes callback costs microseconds: the objects are pooled

    _pool = threading.local()

    @classmethod
    def three(cls, device):
        """(geometry, binning, image) allocators of this thread for `device`, reused from call to call: the forward runs
        on the host's critical path between two GPU phases, so its fixed Python cost matters on a slow host."""
        pool = cls._pool.__dict__.setdefault("by_device", {})
        key = torch.device(device).index
        a = pool.get(key)
        if a is None:
            a = pool[key] = (cls(device), cls(device), cls(device))
        for x in a:
            x.tensor = None
        return a

    def take(self):
        t, self.tensor = self.tensor, None
        return t if t is not None else torch.empty(0, dtype=torch.uint8, device=self.device)

    def _alloc(self, nbytes, _user):
        try:
            self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            return self.tensor.data_ptr()
        except Exception:  # out of memory -> NULL, reported as GS2M_ERR_ALLOC
            return 0


def _stream():
    return _native.stream_ptr()


class _ScratchCache(_Alloc):
    """Backward scratch (partial-gradient rows: ~0.4 GB at 1M Gaussians / 1080p), kept per (device, stream).  The scratch
    is dead once the backward's kernels have run, and calls on one stream are ordered, so the next backward may reuse it;
    handing a block of that size back to the caching allocator every step only invites it to split the block for smaller
    requests and to hipMalloc a new one the step after.  Grown on demand (10 % head room: the row count moves with the
    camera); given back when the last SHRINK_AFTER backwards all asked for less than half of it (evaluation at another
    resolution after training, a pruned model)."""
    SHRINK_AFTER = 32
    _cache = {}
    _cache_lock = threading.Lock()  # autograd runs backwards on its own threads: two devices' backwards may get() at once

    @classmethod
    def get(cls, device, stream):
        key = (torch.device(device).index, stream)
        with cls._cache_lock:
            a = cls._cache.get(key)
            if a is None:
                a = cls._cache[key] = cls(device)
        return a

    @classmethod
    def release(cls):
        """Hand the cached scratch blocks back to PyTorch's allocator (e.g. after training, before evaluation at another
        resolution).  Only call with no backward in flight on the streams concerned."""
        with cls._cache_lock:
            cls._cache.clear()

    recent_max = 0
    window = 0

    def _alloc(self, nbytes, _user):
        nbytes = int(nbytes)
        self.recent_max = max(self.recent_max, nbytes)
        self.window += 1
        if self.window >= self.SHRINK_AFTER:
            if self.tensor is not None and self.tensor.numel() > 2 * self.recent_max + (1 << 20):
                self.tensor = None
            self.recent_max, self.window = nbytes, 0
        if self.tensor is not None and self.tensor.numel() >= nbytes:
            return self.tensor.data_ptr()
        self.tensor = None  # drop the old block before growing
        return super()._alloc(nbytes + nbytes // 10, _user)


# forwards of the autograd path / of them: the cached binning buffer was leased to another graph ("busy": allocated through
# Python inside the GPU-idle window) / was empty or too small ("grown": the same, once per size)
BINNING_CACHE_STATS = {"calls": 0, "busy": 0, "grown": 0}


class _BinningLease:
    """Exclusive use of one cache entry's buffer by ONE forward and the graph node it created.  The cache entry holds the
    lease (strongly); the lease holds a weak reference to the autograd node (`attach`), the node holds the lease.  The
    lease has ended when (a) the node has died, or (b) autograd has freed the node's saved tensors -- a backward without
    retain_graph has run -- which the PUBLIC `ctx.saved_tensors` property reports by raising (the check every second
    `.backward()` on a freed graph trips); a training loop that still holds the previous iteration's outputs when it
    renders again (train.py's `out = render(...)`) keeps the node alive but not its saved tensors.  No finalizer, no
    use-count: rounds 3-4 asked `Tensor._use_count()` (a private counter of C++ references) and cleared the entry from a
    lock-free `__del__` that could run after another thread had taken the entry over."""

    def __init__(self):
        self.node = None      # weakref to the autograd node, set by attach()
        self.ended = False    # set explicitly when the forward that took the lease did not end up using the cached buffer

    def attach(self, ctx):
        import weakref
        self.node = weakref.ref(ctx)

    def over(self):
        """True once nothing can read the leased buffer any more."""
        if self.ended:
            return True
        if self.node is None:
            return False      # the forward that took the lease is still running (or never attached a node): in use
        ctx = self.node()
        if ctx is None:
            return True       # the graph node is gone
        try:
            ctx.saved_tensors  # raises once the saved tensors have been freed (backward ran, graph not retained)
        except RuntimeError:
            return True
        except Exception:
            return False
        return False          # the graph is alive and may still run its backward (again: retain_graph)


class _BinningCache:
    """The binning buffer can only be sized after the forward's host wait for num_rendered, and the GPU idles from that
    wait until the next kernel is launched: a Python allocator callback in that window is pure GPU idle time.  So the
    autograd path keeps one buffer per (device, stream) from call to call (25 % larger than the largest request seen)
    and hands it to the library through its C-level gs2m_prealloc_alloc; Python is only called when it does not fit.
    Ownership is explicit: a forward takes a _BinningLease on the entry and parks it on its autograd node; a forward
    that finds the entry leased -- two views rendered before either backward, a retained graph -- allocates as before.
    Entries are per stream because the buffer bypasses the caching allocator's stream tracking: work queued on one
    stream is ordered, so the next forward on THAT stream may overwrite what the previous backward has finished with.
    Retained memory: 1.25 x the largest binning buffer per (device, stream) until release_scratch(); an entry whose buffer
    is more than twice what the last SHRINK_AFTER forwards asked for (a change of resolution, evaluation after training)
    is given back to the allocator and grown afresh."""
    _cache = {}
    _lock = threading.RLock()
    SHRINK_AFTER = 32

    def __init__(self):
        self.tensor = None
        self.owner = None   # the lease in force, if any
        self.recent_max = 0  # largest request of the current window of SHRINK_AFTER forwards
        self.window = 0

    @classmethod
    def acquire(cls, device, stream):
        """-> (entry, lease); lease is None when the entry is in use (see _BinningLease.over)."""
        key = (torch.device(device).index, stream)
        lease = _BinningLease()
        with cls._lock:
            e = cls._cache.get(key)
            if e is None:
                e = cls._cache[key] = cls()
            if e.owner is not None and not e.owner.over():
                return e, None
            e.owner = lease
        return e, lease

    def note_request(self, nbytes):
        """Shrink policy: called once per forward that holds the lease, with the bytes the library asked for."""
        self.recent_max = max(self.recent_max, int(nbytes))
        self.window += 1
        if self.window >= self.SHRINK_AFTER:
            if self.tensor is not None and self.tensor.numel() > 2 * max(self.recent_max, 1) + 8192:
                self.tensor = None  # regrown (1.25 x the request) by the next forward
            self.recent_max, self.window = 0, 0

    @classmethod
    def release(cls):
        with cls._lock:
            cls._cache.clear()


class _CModule:
    """Function-for-function mirror of the reference's `_C` extension module."""
    _prealloc_cb = None

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                            features, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                            campos, prefiltered, featureCount, sh_rest=None, _cached_binning=False, _lease_out=None):
        """`sh_rest` (this repository's extension): when given, `sh` is the DC part (P,1,3) and `sh_rest` the other
        coefficients (P,M-1,3), as the reference model stores them -- no concatenation needed (M = 16 only)."""
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:52-54
        L = _native.lib()
        means3D = _f32c(means3D, "means3D")
        device = means3D.device
        background = _f32c(background, "bg"); colors = _f32c(colors, "colors_precomp")
        opacities = _f32c(opacities, "opacities"); scales = _f32c(scales, "scales")
        rotations = _f32c(rotations, "rotations"); cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
        features = _f32c(features, "features"); viewmatrix = _f32c(viewmatrix, "viewmatrix")
        projmatrix = _f32c(projmatrix, "projmatrix"); sh = _f32c(sh, "shs"); campos = _f32c(campos, "campos")
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
        split = sh_rest is not None and sh_rest.numel() != 0
        if split:
            sh_rest = _f32c(sh_rest, "shs_rest")
            if sh_rest.data_ptr() % 16:  # e.g. a view into a larger tensor: the kernels stream it as float4
                sh_rest = sh_rest.clone()
            M = 1 + sh_rest.size(1)
        # outputs need no zero fill: the kernels write every element
        out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=device)
        out_buffer = torch.empty((NUM_FEATURES, H, W), dtype=torch.float32, device=device)
        radii = torch.empty((P,), dtype=torch.int32, device=device)
        observe = torch.empty((P,), dtype=torch.int32, device=device)
        geom, binning, img = _Alloc.three(device)
        # the autograd path only (`_C` callers own what they get): the cached buffer of this (device, stream), if free
        cache, lease = _BinningCache.acquire(device, _stream()) if _cached_binning else (None, None)
        pre = None
        if lease is not None and cache.tensor is not None:
            pre = _native.Prealloc(cache.tensor.data_ptr(), cache.tensor.numel(), binning.cb, None, 0, 0)
            if _CModule._prealloc_cb is None:
                _CModule._prealloc_cb = C.cast(L.gs2m_prealloc_alloc, _native.ALLOC_FN)
            bin_cb, bin_user = _CModule._prealloc_cb, C.byref(pre)
        else:
            bin_cb, bin_user = binning.cb, None
        with _native.device_guard(device):
            fwd = L.gs2m_raster_forward_split_sh if split else L.gs2m_raster_forward
            sh_args = (_ptr(sh), _ptr(sh_rest)) if split else (_ptr(sh),)
            rendered = fwd(
                geom.cb, None, bin_cb, bin_user, img.cb, None, P, int(degree), int(M), _ptr(background), W, H,
                _ptr(means3D), *sh_args, _ptr(colors), _ptr(opacities), _ptr(scales), float(scale_modifier),
                _ptr(rotations), _ptr(cov3D_precomp), _ptr(features), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos),
                float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), int(featureCount), _ptr(out_color),
                _ptr(radii), _ptr(observe), _ptr(out_buffer), _stream())
        _native.check(rendered, "gs2m_raster_forward")
        if _cached_binning:
            BINNING_CACHE_STATS["calls"] += 1
            if lease is None:
                BINNING_CACHE_STATS["busy"] += 1
            elif pre is None or pre.used_fallback:
                BINNING_CACHE_STATS["grown"] += 1
        if pre is not None and not pre.used_fallback:
            bin_tensor = cache.tensor
            cache.note_request(pre.requested)
            if _lease_out is not None:
                _lease_out.append(lease)  # the caller attaches its graph node: leased for as long as that node can run a backward
            else:
                lease.ended = True        # nobody to tell us when the buffer is dead: the caller got a cached buffer it must copy or use at once
        else:
            bin_tensor = binning.take()
            if lease is not None:  # the entry is ours but empty or too small: a buffer 25 % larger than this request for the next call
                cache.tensor = torch.empty(int(bin_tensor.numel() * 1.25) + 4096, dtype=torch.uint8, device=device)
                cache.recent_max, cache.window = 0, 0
                lease.ended = True  # this call's buffer is its own allocation: the cached one is free
        return rendered, out_color, radii, observe, out_buffer, geom.take(), bin_tensor, img.take()

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, buffer, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, features, viewmatrix, projmatrix, tan_fovx, tan_fovy, grad_colors,
                                     grad_buffer, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                     featureCount, return_conics=False, sh_rest=None, unused_input_grads=True, want_sh_grad=True):
        """`unused_input_grads=False`: dL/dcolors_precomp and dL/dcov3D_precomp are not computed (None in the result) when
        those inputs were not given -- the reference's extension writes them regardless and its autograd Function drops
        them (diff_gaussian_rasterization/__init__.py:127-139); the Function below asks for this form.  `want_sh_grad=False`:
        dL/dSH is not computed either (None in the result): the caller knows the colour gradient to be identically zero."""
        L = _native.lib()
        device = means3D.device
        means3D = _f32c(means3D, "means3D")
        P = means3D.size(0)
        H, W = grad_colors.size(1), grad_colors.size(2)
        M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
        split = sh_rest is not None and sh_rest.numel() != 0
        if split:
            sh_rest = _f32c(sh_rest, "shs_rest")
            if sh_rest.data_ptr() % 16:
                sh_rest = sh_rest.clone()
            M = 1 + sh_rest.size(1)
        # every tensor goes through .contiguous() as in the forward (and as the reference binding does in both
        # directions, rasterize_points.cu:170-196): the reference Camera builds world_view_transform as
        # torch.tensor(...).transpose(0, 1).cuda(), which is NOT contiguous (scene/cameras.py:64)
        grad_colors = _f32c(grad_colors, "grad_colors"); grad_buffer = _f32c(grad_buffer, "grad_buffer")
        background = _f32c(background, "bg"); colors = _f32c(colors, "colors_precomp")
        scales = _f32c(scales, "scales"); rotations = _f32c(rotations, "rotations")
        cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp"); features = _f32c(features, "features")
        viewmatrix = _f32c(viewmatrix, "viewmatrix"); projmatrix = _f32c(projmatrix, "projmatrix")
        sh = _f32c(sh, "shs"); campos = _f32c(campos, "campos"); buffer = _f32c(buffer, "buffer")
        radii = radii.contiguous()
        # Every gradient tensor is a view into ONE registered arena (gs2m_arena; the kernels write every element, so no
        # zero fill).  What a data-parallel step sums across ranks comes first and adjacent -- means3D, opacities, scales,
        # rotations, features, SH (the SH tensor(s) last, so that a step below the maximal SH degree can sum the rest with
        # one collective and the active bands separately) -- then what it does not: dL/dmeans2D (its per-view NORMS are
        # what densification accumulates, train.py:223-227), and the gradients of inputs that are usually absent
        # (precomputed colours / covariances, scene/gaussian_model.py:230-240 has no such parameter), dL/dconic.
        entries = [("means3D", (P, 3)), ("opacities", (P, 1)), ("scales", (P, 3)), ("rotations", (P, 4)),
                   ("features", (P, NUM_FEATURES))]
        if want_sh_grad:
            entries.append(("shs", (P, 1 if split else M, 3)))
            if split:
                entries.append(("shs_rest", (P, M - 1, 3)))
        want_colors = unused_input_grads or (colors is not None and colors.numel() != 0)
        want_cov3D = unused_input_grads or (cov3D_precomp is not None and cov3D_precomp.numel() != 0)
        entries += [("means2D", (P, 4))] + ([("colors", (P, NUM_CHANNELS))] if want_colors else []) + ([("cov3D", (P, 6))] if want_cov3D else [])
        if return_conics:
            entries.append(("conics", (P, 2, 2)))
        arena = _arena.GradArena(device, entries, zero=(P == 0), key="rasterizer")
        dL_dmeans3D = arena["means3D"]; dL_dmeans2D = arena["means2D"]; dL_dcolors = arena.get("colors")
        dL_dfeatures = arena["features"]; dL_dopacities = arena["opacities"]; dL_dcov3D = arena.get("cov3D")
        dL_dshs = arena.get("shs"); dL_dscales = arena["scales"]; dL_drotations = arena["rotations"]
        dL_dshs_rest = arena.get("shs_rest")
        dL_dconics = arena.get("conics")
        scratch = _ScratchCache.get(device, _stream())
        with _native.device_guard(device):
            bwd = L.gs2m_raster_backward_split_sh if split else L.gs2m_raster_backward
            sh_args = (_ptr(sh), _ptr(sh_rest)) if split else (_ptr(sh),)
            dsh_args = (_ptr(dL_dshs), _ptr(dL_dshs_rest)) if split else (_ptr(dL_dshs),)
            rc = bwd(
                P, int(degree), int(M), int(R), _ptr(background), W, H, _ptr(means3D), *sh_args, _ptr(colors),
                _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(features),
                _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), _ptr(radii),
                _ptr(buffer), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), int(featureCount),
                _ptr(grad_colors), _ptr(grad_buffer), _ptr(dL_dmeans2D), _ptr(dL_dconics), _ptr(dL_dopacities),
                _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), *dsh_args, _ptr(dL_dscales),
                _ptr(dL_drotations), _ptr(dL_dfeatures), scratch.cb, None, _stream())
        _native.check(rc, "gs2m_raster_backward")
        out = (dL_dmeans2D, dL_dcolors, dL_dopacities, dL_dmeans3D, dL_dcov3D, dL_dshs, dL_dscales, dL_drotations,
               dL_dfeatures)
        if split:
            out = out + (dL_dshs_rest,)
        return out + (dL_dconics,) if return_conics else out

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        L = _native.lib()
        means3D = _f32c(means3D, "means3D")
        P = means3D.size(0)
        present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
        if P != 0:
            with _native.device_guard(means3D.device):
                rc = L.gs2m_raster_mark_visible(P, _ptr(means3D), _ptr(_f32c(viewmatrix, "viewmatrix")),
                                                _ptr(_f32c(projmatrix, "projmatrix")), present.data_ptr(), _stream())
            _native.check(rc, "gs2m_raster_mark_visible")
        return present


_C = _CModule()


def release_scratch():
    """Free the backward's cached row scratch (~0.9 GB at 1M Gaussians / 1080p, kept per device and stream between calls)."""
    _ScratchCache.release()
    _BinningCache.release()
    _arena.release()


def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, features,
                        raster_settings, shs_rest=None):
    if shs_rest is None:
        shs_rest = torch.Tensor([])
    return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, features, raster_settings, shs_rest)


_zero_image_cache = {}


def _zero_image(buffer):
    """(3,H,W) zeros on buffer's device, cached per shape (read-only input of the backward kernels: no fill per call)"""
    key = (buffer.device, buffer.shape[1], buffer.shape[2])
    z = _zero_image_cache.get(key)
    if z is None:
        if len(_zero_image_cache) > 8:
            _zero_image_cache.clear()
        z = _zero_image_cache[key] = torch.zeros((NUM_CHANNELS, buffer.shape[1], buffer.shape[2]), dtype=torch.float32, device=buffer.device)
    return z


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, features,
                raster_settings, shs_rest):
        args = (raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, features, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy,
                raster_settings.image_height, raster_settings.image_width, shs, raster_settings.sh_degree,
                raster_settings.campos, raster_settings.prefiltered, raster_settings.feature_count)
        lease = []
        num_rendered, color, radii, observe, buffer, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(
            *args, sh_rest=shs_rest, _cached_binning=True, _lease_out=lease)
        # names this forward's landing slot: the backward sizes its scratch by the dense gradient-row count the GPU leaves there
        ctx.forward_token = _native.lib().gs2m_raster_forward_token()
        ctx.binning_lease = lease[0] if lease else None
        if lease:
            lease[0].attach(ctx)  # over once this node has died or autograd has freed its saved tensors (_BinningLease.over)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        # radii and observe carry no gradient: without this autograd materialises a zero tensor for each of them in front
        # of every backward (two fill kernels of P elements on the stream)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(buffer, features, colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, shs,
                              geomBuffer, binningBuffer, imgBuffer, shs_rest)
        ctx.mark_non_differentiable(radii, observe)
        return color, radii, observe, buffer

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_radii, grad_out_observe, grad_out_buffer):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (buffer, features, colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, shs, geomBuffer,
         binningBuffer, imgBuffer, shs_rest) = ctx.saved_tensors
        # No gradient reached the colour output (materialize_grads is off): a view rendered for its G-buffer only -- the multi-view
        # term's neighbour view (train.py:121-130 uses its depth and normal maps).  dL/dcolour is then identically zero for every
        # Gaussian, and with it dL/dSH: the kernel is told not to write the (P,16,3) tensor, and autograd gets None (no zeros to add
        # to the other view's SH gradient: 0.06 ms of accumulation per training iteration at 590 k Gaussians).
        color_dead = grad_out_color is None and os.environ.get("GS2M_KEEP_DEAD_SH") is None  # (env: A/B of this shortcut)
        if grad_out_color is None:
            grad_out_color = _zero_image(buffer)
        if grad_out_buffer is None:
            grad_out_buffer = torch.zeros_like(buffer)
        args = (raster_settings.bg, means3D, radii, buffer, colors_precomp, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, features, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy, grad_out_color,
                grad_out_buffer, shs, raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered,
                binningBuffer, imgBuffer, raster_settings.feature_count)
        L = _native.lib()
        dense_rows = L.gs2m_raster_dense_rows(ctx.forward_token)  # -1: no longer available (worst-case sizing: 4 rows per instance)
        if dense_rows >= 0:
            L.gs2m_raster_backward_rows_hint(dense_rows)  # consumed by this thread's next backward call, i.e. the one below
        res = _C.rasterize_gaussians_backward(*args, sh_rest=shs_rest, unused_input_grads=False, want_sh_grad=not color_dead)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_features) = res[:9]
        grad_sh_rest = res[9] if len(res) > 9 else None

        def opt(g, x):  # gradients for absent optionals (empty placeholder tensors) are dropped
            return g if (x is not None and x.numel() != 0) else None

        return (grad_means3D, grad_means2D, opt(grad_sh, shs), opt(grad_colors_precomp, colors_precomp), grad_opacities,
                opt(grad_scales, scales), opt(grad_rotations, rotations), opt(grad_cov3Ds_precomp, cov3Ds_precomp),
                opt(grad_features, features), None, grad_sh_rest)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    feature_count: int


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, features=None, shs_rest=None):
        """`shs_rest` is this repository's extension: with it, `shs` is the DC coefficient (P,1,3) and `shs_rest` the
        others (P,15,3) -- the two tensors the reference model keeps -- and no concatenated copy is needed."""
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        if features is None:
            features = torch.Tensor([])

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   features, raster_settings, shs_rest=shs_rest)
