"""Gradient arenas: several gradient tensors allocated inside ONE flat fp32 buffer, with an explicit registry.

Data-parallel training sums per-Gaussian gradients across ranks at step end (gs2m_dp).  Tensors that an
op's backward allocates side by side in one arena can be summed IN PLACE with one collective over the exact range they
occupy -- no concatenated copy.  The registry is what makes that safe: the reducer only ever touches a range whose
every entry it was handed (gs2m_dp.GradReducer.reduce_flat), never "whatever else lives in that storage".
Producers: the rasterizer binding's backward (diff_gaussian_rasterization) and the fused activation backward
(gs2m_render_ops._Activate)."""
import collections
import threading

import torch

_ALIGN = 4  # floats: entries start on 16-B boundaries (the kernels stream SH rows as float4)
_KEEP = 2   # arenas kept registered (and thereby alive) per producer and device
_lock = threading.Lock()
_registry = {}  # (producer key, device index) -> deque of the producer's latest GradArenas
_enabled = False  # arenas are registered (and thereby kept alive) only once a reducer exists: enable()


def _dist_up():
    """a process group exists: a reducer may be created at any moment (and sum gradients that already exist)"""
    d = torch.distributed
    return d.is_available() and d.is_initialized()


def set_keep(n):
    """Arenas kept registered per producer and device (default 2: the current step's and the previous one's).  A step that
    ACCUMULATES the gradients of V views in the first view's arena (train(views_per_rank=V), bench.py --views-per-rank V)
    produces V arenas before it reduces the first: it asks for V + 1, so that the first is still registered -- hence still
    summed in place -- when the reduction runs."""
    global _KEEP
    n = max(2, int(n))
    with _lock:
        if n != _KEEP:
            _KEEP = n
            for k in list(_registry):
                _registry[k] = collections.deque(_registry[k], maxlen=n)


def enable(on=True):
    """Start (stop) registering new arenas.  Called by gs2m_dp.GradReducer: only a reducer ever looks an arena up, and a
    registered arena outlives its gradients (`_KEEP` per producer: ~0.65 GB at 1M Gaussians for the rasterizer's), which
    single-GPU training should not pay for."""
    global _enabled
    _enabled = bool(on)
    if not on:
        release()


class GradArena:
    """entries: list of (name, shape).  `self[name]` is the view; `self.layout` = [(name, offset, numel, shape)] in floats, in
    the order given -- callers put what a data-parallel step sums FIRST / ADJACENT so that it forms one contiguous range.
    Once enable() has been called (a data-parallel reducer exists) the registry holds the `_KEEP` latest arenas of every producer (`key`) per device: a registered arena's buffer is
    alive, so a tensor whose storage starts where the arena's does IS one of its views (an address cannot have been
    reused).  Older arenas drop out -- their gradients are then summed through a copy -- and gs2m_arena.release() (also
    called by diff_gaussian_rasterization.release_scratch) lets go of all of them.  Retained memory: at most `_KEEP`
    arenas per producer beyond the gradients' own lifetime (the rasterizer's: 82 floats per Gaussian)."""

    def __init__(self, device, entries, zero=False, key="default"):
        self.layout, total = [], 0
        for name, shape in entries:
            n = 1
            for d in shape:
                n *= int(d)
            self.layout.append((name, total, n, tuple(int(d) for d in shape)))
            total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = (torch.zeros if zero else torch.empty)(max(total, 1), dtype=torch.float32, device=device)
        # no view is kept here: autograd takes a gradient over as a leaf's .grad WITHOUT a copy only while nothing else
        # references the tensor (AccumulateGrad checks its use count) -- a registry that held the views made every
        # backward clone all of its gradients
        self._index = {name: (off, n, shape) for name, off, n, shape in self.layout}
        self.ptr = self.flat.untyped_storage().data_ptr()
        if _enabled or _dist_up():
            with _lock:
                _registry.setdefault((key, self.flat.device.index), collections.deque(maxlen=_KEEP)).append(self)

    def __getitem__(self, name):
        off, n, shape = self._index[name]
        return self.flat[off:off + n].view(shape)

    def get(self, name):
        return self[name] if name in self._index else None


def release():
    with _lock:
        _registry.clear()


def lookup(t):
    """-> (arena, offset, numel) when `t` is exactly one entry of a registered arena, else None."""
    if t is None or t.dtype != torch.float32 or not t.is_contiguous():
        return None
    ptr = t.untyped_storage().data_ptr()
    with _lock:
        arena = next((a for dq in _registry.values() for a in dq if a.ptr == ptr), None)
    if arena is None:
        return None
    off = t.storage_offset()
    for name, o, n, shape in arena.layout:
        if o == off and n == t.numel():
            return arena, o, n
    return None


def contiguous_range(arena, items):
    """items: [(offset, numel)] of entries of `arena`.  -> (start, end) of the smallest range covering them if that range
    holds NO entry of the arena that is not among them (alignment padding is allowed), else None."""
    have = {o for o, n in items}
    start = min(o for o, n in items)
    end = max(o + n for o, n in items)
    for name, o, n, shape in arena.layout:
        if n > 0 and o < end and o + n > start and o not in have:
            return None
    return start, end
