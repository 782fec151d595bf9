"""Gradient arenas: several gradient tensors allocated inside ONE flat fp32 buffer, with an explicit registry.

Data-parallel training sums per-Gaussian gradients across ranks at step end (gs2m_dp).  Tensors that an
op's backward allocates side by side in one arena can be summed IN PLACE with one collective over the exact range they
occupy -- no concatenated copy.  The registry is what makes that safe: the reducer only ever touches a range whose
every entry it was handed (gs2m_dp.GradReducer.reduce_flat), never "whatever else lives in that storage".
Producers: the rasterizer binding's backward (diff_gaussian_rasterization) and the fused activation backward
(gs2m_render_ops._Activate)."""
import threading
import weakref

import torch

_ALIGN = 4  # floats: entries start on 16-B boundaries (the kernels stream SH rows as float4)
_lock = threading.Lock()
_registry = {}  # storage data_ptr -> weakref to the GradArena that owns it


class GradArena:
    """entries: list of (name, shape).  `self[name]` is the view; `self.layout` = [(name, offset, numel)] in floats, in
    the order given -- callers put what a data-parallel step sums FIRST / ADJACENT so that it forms one contiguous range."""

    def __init__(self, device, entries, zero=False):
        self.layout, total = [], 0
        for name, shape in entries:
            n = 1
            for d in shape:
                n *= int(d)
            self.layout.append((name, total, n, tuple(int(d) for d in shape)))
            total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = (torch.zeros if zero else torch.empty)(max(total, 1), dtype=torch.float32, device=device)
        self._views = {name: self.flat[off:off + n].view(shape) for name, off, n, shape in self.layout}
        with _lock:
            for k in [k for k, r in _registry.items() if r() is None]:
                del _registry[k]
            _registry[self.flat.untyped_storage().data_ptr()] = weakref.ref(self)

    def __getitem__(self, name):
        return self._views[name]

    def get(self, name):
        return self._views.get(name)


def lookup(t):
    """-> (arena, offset, numel) when `t` is exactly one registered entry of a live arena, else None."""
    if t is None or t.dtype != torch.float32 or not t.is_contiguous():
        return None
    with _lock:
        ref = _registry.get(t.untyped_storage().data_ptr())
    arena = ref() if ref is not None else None
    if arena is None or arena.flat.untyped_storage().data_ptr() != t.untyped_storage().data_ptr():
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
