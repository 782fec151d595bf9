"""View-parallel data parallelism for the rasterizer hot path.

The reference has no multi-GPU code.  The path shards naturally by view: one process per
GPU, every rank holds the full (replicated) Gaussian parameters, renders its own camera,
and the per-Gaussian gradients are summed at step end with ONE collective pass over RCCL
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  There is no
exchange inside forward/backward.

Dense gradient payload per Gaussian (param list scene/gaussian_model.py:230-240):
xyz 3, SH 3*M, opacity 1, scaling 3, rotation 4, material (albedo 3, roughness 1,
metallic 1) = 64 floats at M = 16.  Side channels used by densification
(train.py:225-227, scene/gaussian_model.py:569-573) are reduced with the semantics the
single-GPU loop has: per-view norms of the screen-space gradient are summed, the
visibility count is summed, radii are max-reduced and `observe` is summed.
"""
import torch
import torch.distributed as dist


class PendingReduce:
    """Collectives in flight (GradReducer.reduce_*_async).  Holds the tensors alive until `wait()`."""

    def __init__(self, result):
        self.result = result
        self.handles = []
        self.restore = []
        self.finish = None

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        for g, part in self.restore:
            g[:, :part.shape[1]] = part
        self.restore = []
        if self.finish is not None:
            self.result = self.finish()
            self.finish = None
        return self.result


class GradReducer:
    """Sums per-Gaussian gradient tensors across ranks, in place.

    mode "allreduce": one async all-reduce per tensor (RCCL picks the algorithm).
    mode "rs_ag": reduce-scatter + all-gather on a flat view, which keeps every xGMI link of
    the fully connected 8-GPU mesh busy instead of a single-link-bound ring; needs numel
    divisible by world size (tensors are padded internally otherwise fall back to allreduce).
    """

    def __init__(self, group=None, mode="allreduce", sh_active_coeffs=None):
        self.group = group
        self.mode = mode
        self.sh_active_coeffs = sh_active_coeffs

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _sum(self, t, handles):
        ws = self.world_size
        flat = t.view(-1)
        if self.mode == "rs_ag" and flat.numel() % ws == 0 and flat.numel() >= ws:
            shard = torch.empty(flat.numel() // ws, dtype=flat.dtype, device=flat.device)
            h = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            # RCCL runs a group's collectives in issue order on its own stream: the all-gather can be queued behind the
            # reduce-scatter without the host waiting in between.  Other backends (gloo in the CPU tests) give no such
            # ordering: wait there.
            if dist.get_backend(self.group) != "nccl":
                h.wait()
            else:
                handles.append(h)
            handles.append(dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True))
        else:
            handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_grads_async(self, grads):
        """Starts the sum of `grads` (dict name -> contiguous tensor, summed in place across ranks) and returns a
        PendingReduce; nothing waits yet.  The collectives run on the backend's own stream behind the work already
        queued on the current stream, so whatever the caller launches next -- the NEXT view's forward and backward
        -- overlaps them; `.wait()` then makes the current stream (gloo: the host) wait and returns the dict.
        The tensors must not be touched before `.wait()`."""
        pend = PendingReduce(grads)
        if self.world_size == 1:
            return pend
        for name, g in grads.items():
            if g is None:
                continue
            if name == "shs" and self.sh_active_coeffs is not None and self.sh_active_coeffs < g.shape[1]:
                # bands above the active degree have exactly zero gradient on every rank
                part = g[:, :self.sh_active_coeffs].contiguous()
                self._sum(part, pend.handles)
                pend.restore.append((g, part))
            else:
                assert g.is_contiguous(), name
                self._sum(g, pend.handles)
        return pend

    def reduce_grads(self, grads):
        """grads: dict name -> contiguous tensor (summed in place across ranks). Returns the dict."""
        return self.reduce_grads_async(grads).wait()

    # ---- one collective per step --------------------------------------------------------------------------------
    @staticmethod
    def common_arena(tensors):
        """The flat fp32 tensor over the storage the given tensors all live in, or None.  The rasterizer binding
        allocates every gradient it returns inside ONE buffer (diff_gaussian_rasterization: _grad_arena), so the
        nine per-Gaussian gradients of a view can be summed with a single collective and no copy."""
        ts = [t for t in tensors if t is not None and t.numel() > 0]
        if not ts or any(t.dtype != torch.float32 or not t.is_contiguous() for t in ts):
            return None
        st = ts[0].untyped_storage()
        if any(t.untyped_storage().data_ptr() != st.data_ptr() for t in ts[1:]):
            return None
        covered = sum(t.numel() for t in ts)
        total = st.nbytes() // 4
        if covered * 2 < total:  # mostly something else's memory (e.g. slices of a large parameter blob)
            return None
        return torch.empty(0, dtype=torch.float32, device=ts[0].device).set_(st, 0, (total,))

    def reduce_flat_async(self, tensors):
        """Sum of a LIST of gradient tensors across ranks with ONE collective: in place on their common arena when they
        share one, otherwise through one concatenated copy (the returned tensors are then views of that copy).
        -> PendingReduce whose .wait() gives the list of reduced tensors (same order; None entries stay None)."""
        live = [t for t in tensors if t is not None]
        pend = PendingReduce(list(tensors))
        if self.world_size == 1 or not live:
            return pend
        arena = self.common_arena(live)
        if arena is not None:
            self._sum(arena, pend.handles)
            return pend
        flat = torch.cat([t.reshape(-1) for t in live])
        pad = (-flat.numel()) % self.world_size
        if pad:
            flat = torch.cat([flat, flat.new_zeros(pad)])
        self._sum(flat, pend.handles)
        out, off = [], 0
        for t in tensors:
            if t is None:
                out.append(None)
            else:
                out.append(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()
        pend.result = out
        return pend

    def reduce_flat(self, tensors):
        return self.reduce_flat_async(tensors).wait()

    def reduce_parameter_grads(self, params):
        """`p.grad` of every parameter <- sum over ranks, one collective (the parameters keep views of one flat buffer
        as their .grad, which is what the optimizer then reads)."""
        params = [p for p in params if p.grad is not None]
        red = self.reduce_flat([p.grad for p in params])
        for p, g in zip(params, red):
            if g is not p.grad:
                p.grad = g

    def reduce_densification_stats_async(self, viewspace_grad, radii, observe):
        """Per-view statistics -> what a single process would have accumulated over all ranks' views; returns a
        PendingReduce whose `.wait()` gives (grad_norm_sum (P,1), grad_abs_norm_sum (P,1), visible_count (P,1),
        max_radii (P), observe_sum (P))."""
        vis = (radii > 0)
        gn = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
        ga = torch.norm(viewspace_grad[:, 2:], dim=-1, keepdim=True) * vis[:, None]
        packed = torch.cat([gn, ga, vis[:, None].to(gn.dtype), observe[:, None].to(gn.dtype)], dim=1).contiguous()
        mr = radii.clone()
        pend = PendingReduce(None)
        if self.world_size > 1:
            pend.handles.append(dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            pend.handles.append(dist.all_reduce(mr, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        pend.finish = lambda: (packed[:, 0:1], packed[:, 1:2], packed[:, 2:3], mr, packed[:, 3].round().to(observe.dtype))
        return pend

    def reduce_densification_stats(self, viewspace_grad, radii, observe):
        return self.reduce_densification_stats_async(viewspace_grad, radii, observe).wait()


def shard_views(num_views, rank, world_size):
    """Views rank r renders in one step: r, r + world, ...."""
    return list(range(rank, num_views, world_size))
