"""world_size-2 gloo test of the view-parallel gradient reduction (gs2m_dp).  The per-rank
rasterizer gradients come from the CPU oracle here (test infrastructure; on the GPU box the HIP op
produces them), so what is under test is the collective logic: the reduced gradients and
densification statistics on every rank equal what one process rendering both views accumulates."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _grads_for_view(view, P=600, W=64, H=48):
    for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gs2m_synth as S
    import helpers as Hh
    from oracle import oracle
    cams = S.orbit_cameras(4, W, H, radius=1.5, centre=(0.0, 0.0, 6.0))
    sc = Hh.make_scene(P, W, H, seed=21, fc=9, scale_hi=0.08, cam=cams[view])
    sc["g"] = S.make_gaussians(P, S.make_camera(W, H), seed=21, scale_hi=0.08)
    f, gr = Hh.run_oracle(oracle, sc)
    g = {k: torch.tensor(np.ascontiguousarray(gr[k])) for k in ("means3D", "shs", "opacities", "scales", "rotations", "features")}
    return g, torch.tensor(gr["means2D"]), torch.tensor(f.radii), torch.tensor(f.observe)


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
    from gs2m_dp import GradReducer, shard_views
    assert shard_views(4, rank, world) == [rank, rank + 2]
    g, m2d, radii, observe = _grads_for_view(rank)
    red = GradReducer(mode=mode, sh_active_coeffs=16)
    if mode == "allreduce":  # blocking form
        red.reduce_grads(g)
        stats = red.reduce_densification_stats(m2d, radii, observe)
    else:  # pipelined form (bench.py at N > 1): both reductions in flight, other work in between, then wait
        p1 = red.reduce_grads_async(g)
        p2 = red.reduce_densification_stats_async(m2d, radii, observe)
        _ = torch.ones(1000).sum()  # stands for the next view's forward + backward
        assert p1.wait() is g
        stats = p2.wait()
    q.put((rank, {k: v.numpy() for k, v in g.items()}, [s.numpy() for s in stats]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_two_rank_gradient_sum(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "allreduce" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, m0, r0, o0 = _grads_for_view(0)
    g1, m1, r1, o1 = _grads_for_view(1)
    for rank, g, stats in res:
        for k in g0:
            assert np.allclose(g[k], (g0[k] + g1[k]).numpy(), rtol=1e-6, atol=1e-7), (rank, k)
        v0, v1 = (r0 > 0).float()[:, None], (r1 > 0).float()[:, None]
        gn = torch.norm(m0[:, :2], dim=-1, keepdim=True) * v0 + torch.norm(m1[:, :2], dim=-1, keepdim=True) * v1
        ga = torch.norm(m0[:, 2:], dim=-1, keepdim=True) * v0 + torch.norm(m1[:, 2:], dim=-1, keepdim=True) * v1
        assert np.allclose(stats[0], gn.numpy(), rtol=1e-6) and np.allclose(stats[1], ga.numpy(), rtol=1e-6)
        assert np.array_equal(stats[2], (v0 + v1).numpy())
        assert np.array_equal(stats[3], torch.maximum(r0, r1).numpy())
        assert np.array_equal(stats[4], (o0 + o1).numpy())
