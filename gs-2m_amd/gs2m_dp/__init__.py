"""View-parallel data parallelism for the rasterizer hot path.

The reference has no multi-GPU code.  The path shards naturally by view: one process per
GPU, every rank holds the full (replicated) Gaussian parameters, renders its own camera,
and the per-Gaussian gradients are summed at step end with ONE collective pass over RCCL
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  There is no
exchange inside forward/backward.

Dense gradient payload per Gaussian (param list scene/gaussian_model.py:230-240):
xyz 3, SH 3*M, opacity 1, scaling 3, rotation 4, material (albedo 3, roughness 1,
metallic 1) = 64 floats at M = 16; below the maximal SH degree only the active bands
travel (19 / 28 / 43 / 64 floats at degree 0 / 1 / 2 / 3).  The sums run in place over
registered gradient arenas (gs2m_arena) wherever autograd left the gradients in one.  Side channels used by densification
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
        for g, part in self.restore:  # SH gradients summed as a packed copy of their active bands
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

    def __init__(self, group=None, mode="allreduce", sh_active_coeffs=None, always_communicate=False):
        self.group = group
        self.mode = mode
        self.sh_active_coeffs = sh_active_coeffs
        self.always_communicate = always_communicate  # issue the collectives at world size 1 too (tests of the RCCL path on one GPU)
        self.last_plan = []
        import gs2m_arena
        gs2m_arena.enable()  # from now on the producers' gradient arenas are registered (summed in place)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            try:  # the collectives' kernels share the device with the rasterizer from now on: the tile sort must not assume
                import gs2m_native  # that every workgroup of a pass is resident at once (radix_sort.hip: launch_pass)
                import torch
                if torch.cuda.is_available():
                    gs2m_native.set_sort_tickets(True)
            except Exception:
                pass

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
        if self.world_size == 1 and not self.always_communicate:
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

    # ---- arenas: in-place sums over exactly the ranges the caller owns ----------------------------------------------
    def reduce_flat_async(self, tensors, sh_active=None):
        """Sum of a LIST of gradient tensors across ranks.  Tensors that are entries of a registered gradient arena
        (gs2m_arena: the rasterizer binding's backward, the fused activation backward) are summed IN PLACE, one
        collective per arena, over the smallest range that covers them -- and only if that range holds nothing the
        caller did not pass (otherwise, and for loose tensors, one concatenated copy is summed and views of it are
        returned).  Nothing outside the passed tensors is ever modified.
        `sh_active`: {index into `tensors`: n} for SH gradients (P, M, 3) of which only the first n coefficients can be
        non-zero on any rank (train.py:81-82: the SH degree is raised every 1000 iterations; bands above it receive no
        gradient): only those travel, as one packed copy (n = 0: nothing travels); the binding puts the SH gradients at
        the end of its arena's summed range so that the rest stays one range.
        -> PendingReduce whose .wait() gives the list of reduced tensors (same order; None entries stay None)."""
        import gs2m_arena
        pend = PendingReduce(list(tensors))
        live = [(i, t) for i, t in enumerate(tensors) if t is not None and t.numel() > 0]
        if (self.world_size == 1 and not self.always_communicate) or not live:
            return pend
        trimmed = {}
        for i, n in (sh_active or {}).items():
            t = tensors[i]
            if t is not None and t.dim() == 3 and 0 <= n < t.shape[1]:
                trimmed[i] = t[:, :n].contiguous()
        groups, loose = {}, []
        for i, t in live:
            if i in trimmed:
                continue
            info = gs2m_arena.lookup(t)
            if info is None:
                loose.append((i, t))
            else:
                groups.setdefault(id(info[0]), (info[0], []))[1].append((i, t, info[1], info[2]))
        self.last_plan = []  # what the call did, for tests: [("arena", numel) | ("copy", numel) | ("sh", numel)]
        for arena, items in groups.values():
            rng = gs2m_arena.contiguous_range(arena, [(o, n) for _, _, o, n in items])
            if rng is None:
                loose += [(i, t) for i, t, _, _ in items]
                continue
            self._sum(arena.flat[rng[0]:rng[1]], pend.handles)
            self.last_plan.append(("arena", rng[1] - rng[0]))
        for i, part in trimmed.items():
            if part.numel():
                self._sum(part, pend.handles)
                pend.restore.append((tensors[i], part))
            self.last_plan.append(("sh", part.numel()))
        if loose:
            flat = torch.cat([t.reshape(-1) for _, t in loose])
            pad = (-flat.numel()) % max(self.world_size, 1)
            if pad:
                flat = torch.cat([flat, flat.new_zeros(pad)])
            self._sum(flat, pend.handles)
            self.last_plan.append(("copy", flat.numel()))
            off = 0
            for i, t in loose:
                pend.result[i] = flat[off:off + t.numel()].view(t.shape)
                off += t.numel()
        return pend

    def reduce_flat(self, tensors, sh_active=None):
        return self.reduce_flat_async(tensors, sh_active).wait()

    def reduce_parameter_grads(self, params, sh_active=None):
        """`p.grad` of every parameter <- sum over ranks.  Gradients that autograd left inside a registered arena (the SH
        gradients straight from the rasterizer, the six raw-parameter gradients of the fused activation backward) are
        summed where they are; the rest (dL/dxyz: a sum of several contributions) through one small concatenated copy
        whose views become the .grad.  `sh_active`: [(parameter, n)] for SH parameters of which only the first n
        coefficients have a gradient (band trimming, see reduce_flat_async)."""
        params = [p for p in params if p.grad is not None]
        sh = {k: n for k, p in enumerate(params) for q, n in (sh_active or []) if p is q}
        red = self.reduce_flat([p.grad for p in params], sh_active=sh)
        for p, g in zip(params, red):
            if g is not p.grad:
                p.grad = g

    @staticmethod
    def local_densification_stats(viewspace_grad, radii, observe, into=None):
        """One view's densification statistics as (packed (P,4): gradient norm, |.|-gradient norm, visible, observe; radii
        (P)), added to `into` (the same pair from the rank's earlier views of this step: sums, and the max of the radii) when
        given -- the accumulate mode (several views per rank and step) reduces them ONCE, after the last view."""
        vis = (radii > 0)
        gn = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
        ga = torch.norm(viewspace_grad[:, 2:], dim=-1, keepdim=True) * vis[:, None]
        packed = torch.cat([gn, ga, vis[:, None].to(gn.dtype), observe[:, None].to(gn.dtype)], dim=1).contiguous()
        if into is None:
            return packed, radii.clone()
        into[0].add_(packed)
        torch.maximum(into[1], radii, out=into[1])
        return into

    def reduce_densification_stats_async(self, viewspace_grad, radii, observe, local=None):
        """Per-view statistics -> what a single process would have accumulated over all ranks' views; returns a
        PendingReduce whose `.wait()` gives (grad_norm_sum (P,1), grad_abs_norm_sum (P,1), visible_count (P,1),
        max_radii (P), observe_sum (P)).  `local`: the rank's views already added up (local_densification_stats)."""
        packed, mr = local if local is not None else self.local_densification_stats(viewspace_grad, radii, observe)
        pend = PendingReduce(None)
        if self.world_size > 1:
            pend.handles.append(dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            pend.handles.append(dist.all_reduce(mr, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        pend.finish = lambda: (packed[:, 0:1], packed[:, 1:2], packed[:, 2:3], mr, packed[:, 3].round().to(torch.int32 if observe is None else observe.dtype))
        return pend

    def reduce_densification_stats(self, viewspace_grad, radii, observe):
        return self.reduce_densification_stats_async(viewspace_grad, radii, observe).wait()


def shard_views(num_views, rank, world_size):
    """Views rank r renders in one step: r, r + world, ...."""
    return list(range(rank, num_views, world_size))
