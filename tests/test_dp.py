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


# ---------------------------------------------------------------------------------------------------------------------
# the training loop under data parallelism: replicas must stay bit-identical
def _train_worker(rank, world, port, q, views_per_rank=1, iterations=12):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import gaussian_renderer
    import gs2m_synth as S
    import gs2m_train
    import helpers as Hh
    import simple_knn._C as knn
    from gs2m_model import OptimizationParams
    from gs2m_scene import Camera, PipelineParams
    from oracle import oracle
    # CPU stand-ins for the device pieces (test infrastructure): the oracle-backed rasterizer behind the same render()
    # code, the oracle's distCUDA2, torch's Adam, and a plain differentiable image similarity in place of fused SSIM --
    # what is under test is the sharding of the loop, not the kernels
    gaussian_renderer.GaussianRasterizer = Hh.oracle_rasterizer_class(oracle)
    knn.distCUDA2 = lambda pts: torch.tensor(oracle.knn_dist2(pts.detach().cpu().numpy()))
    ssim = lambda a, b: 1.0 - ((a - b) ** 2).mean()
    pipe = PipelineParams()
    pipe.fused_render_ops = False
    W, H = 48, 32
    cams = [Camera(c, "cpu") for c in S.orbit_cameras(4, W, H, radius=2.0, centre=(0.0, 0.0, 6.0), fx=1.2 * W)]
    g = torch.Generator().manual_seed(5)
    gts = [torch.rand(3, H, W, generator=g) for _ in cams]
    pts = (torch.rand(260, 3, generator=g) - 0.5) * torch.tensor([1.6, 1.0, 1.0]) + torch.tensor([0.0, 0.0, 6.0])
    cols = torch.rand(260, 3, generator=g)
    scene = (cams, gts, pts.numpy(), cols.numpy(), 3.0)

    class Opt(OptimizationParams):
        densify_from_iter = 2
        densification_interval = 3
        opacity_reset_interval = 6
        densify_until_iter = 11
        densify_grad_threshold = 1e-7      # so that clone AND split both fire on this tiny problem
        densify_grad_abs_threshold = 1e-7
        percent_dense = 0.012
    model, st = gs2m_train.train(iterations=iterations, W=W, H=H, scene=scene, device="cpu", opt=Opt(), dp=world > 1, ssim_fn=ssim,
                                 optimizer_cls=torch.optim.Adam, pipe=pipe, trim_interval=4, geometry_from_iter=7, seed=3,
                                 views_per_rank=views_per_rank)
    state = {}
    for grp in model.optimizer.param_groups:
        p = grp["params"][0]
        sd = model.optimizer.state.get(p, {})
        state[grp["name"]] = (p.detach().numpy().copy(), sd.get("exp_avg", torch.zeros(0)).numpy().copy(),
                              sd.get("exp_avg_sq", torch.zeros(0)).numpy().copy())
    q.put((rank, state, model.max_radii2D.numpy().copy(), model.denom.numpy().copy(), st["points_start"], st["points_end"],
           st.get("trimmed", 0)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_training_replicas_stay_bit_identical():
    """gs2m_train.train(dp=True) on two gloo ranks, 12 iterations spanning densify (clone + split, with the shared
    generator behind torch.normal), prune, the multi-view observe trim and an opacity reset: both ranks must end with
    bit-identical parameters, Adam moments, statistics and point count (train.py:223-254, scene/gaussian_model.py:489-573
    applied to all-reduced side channels)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    res = []
    for _ in range(600):  # up to 10 minutes, but fail at once when a worker has died
        try:
            res.append(q.get(timeout=1.0))
        except _queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        if len(res) == 2:
            break
    assert len(res) == 2
    res.sort(key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, mr0, dn0, n0a, n0b, t0), (_, s1, mr1, dn1, n1a, n1b, t1) = res
    assert n0b == n1b and n0a == n1a and t0 == t1
    assert n0b != n0a, "the run should change the point count (densify / prune / trim)"
    for name in s0:
        for a, b, what in zip(s0[name], s1[name], ("param", "exp_avg", "exp_avg_sq")):
            assert a.shape == b.shape and np.array_equal(a, b), (name, what)
    assert np.array_equal(mr0, mr1) and np.array_equal(dn0, dn1)


def _run_train(world, views_per_rank, iterations, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, views_per_rank, iterations)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    res = []
    for _ in range(600):
        try:
            res.append(q.get(timeout=1.0))
        except _queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        if len(res) == world:
            break
    assert len(res) == world
    res.sort(key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_accumulate_mode_replicas_stay_bit_identical_and_match_one_process():
    """gs2m_train.train(dp=True, views_per_rank=2): every rank renders TWO views per iteration, their gradients accumulate,
    ONE reduction follows the second view, one optimizer step.  (a) 12 iterations with densify / prune / trim / reset on two
    gloo ranks: replicas bit-identical; (b) the gradients are exact sums, not a step late: two iterations (before any
    densification) equal what ONE process accumulating the same four views per iteration produces, up to the association of
    the fp32 sum ((v0 + v1) + (v2 + v3) against ((v0 + v1) + v2) + v3)."""
    port = 35500 + (os.getpid() % 2000)
    res = _run_train(2, 2, 12, port)
    (_, s0, mr0, dn0, n0a, n0b, t0), (_, s1, mr1, dn1, n1a, n1b, t1) = res
    assert n0b == n1b and n0a == n1a and t0 == t1 and n0b != n0a
    for name in s0:
        for a, b, what in zip(s0[name], s1[name], ("param", "exp_avg", "exp_avg_sq")):
            assert a.shape == b.shape and np.array_equal(a, b), (name, what)
    assert np.array_equal(mr0, mr1) and np.array_equal(dn0, dn1)
    two = _run_train(2, 2, 2, port + 1)[0]
    one = _run_train(1, 4, 2, port + 2)[0]
    for name in two[1]:
        for a, b, what in zip(two[1][name], one[1][name], ("param", "exp_avg", "exp_avg_sq")):
            assert a.shape == b.shape and np.allclose(a, b, rtol=2e-4, atol=1e-7), (name, what, float(np.abs(a - b).max()))
    assert np.array_equal(two[3], one[3]), "visibility counts (denom) are integers: identical"


def _acc_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
    from gs2m_dp import GradReducer
    ga = _grads_for_view(2 * rank)[0]
    gb = _grads_for_view(2 * rank + 1)[0]
    for k in ga:  # what autograd's accumulation does with the second view's backward
        ga[k] += gb[k]
    GradReducer(mode="allreduce").reduce_grads(ga)
    q.put((rank, {k: v.numpy() for k, v in ga.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_accumulated_gradients_equal_the_sequential_sum_bit_for_bit():
    """two ranks x two accumulated views, one all-reduce: every rank holds (g0 + g1) + (g2 + g3) BIT FOR BIT -- the sum a
    single process forms when it adds the views' gradients in that order (no rounding is introduced by the collective)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_acc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = [_grads_for_view(v)[0] for v in range(4)]
    for rank, got in res:
        for k in g[0]:
            exp = (g[0][k] + g[1][k]) + (g[2][k] + g[3][k])
            assert np.array_equal(got[k], exp.numpy()), (rank, k)


# ---------------------------------------------------------------------------------------------------------------------
# one collective per step: tensors that share one arena, and the concatenating fallback
def _flat_worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
    from gs2m_dp import GradReducer
    g = torch.Generator().manual_seed(100 + rank)
    red = GradReducer(mode=mode)
    # (a) entries of ONE registered arena, as the rasterizer binding returns its gradients; the summed ones first and
    # adjacent, then two the caller does not pass (they must come back untouched)
    import gs2m_arena
    ar = gs2m_arena.GradArena("cpu", [("a", (10, 3)), ("b", (10, 4)), ("sh", (10, 16, 3)), ("x", (10, 4)), ("y", (10, 6))], zero=True)
    views = [ar["a"], ar["b"], ar["sh"]]
    for name in ("a", "b", "sh", "x", "y"):
        ar[name].copy_(torch.randn(ar[name].shape, generator=g))
    keep_x, keep_y = ar["x"].clone(), ar["y"].clone()
    assert gs2m_arena.lookup(views[0])[0] is ar and gs2m_arena.lookup(ar["sh"][:, :4]) is None
    calls = {"n": 0}
    orig_ar, orig_rs = dist.all_reduce, dist.reduce_scatter_tensor

    def count_ar(*a, **k):
        calls["n"] += 1
        return orig_ar(*a, **k)

    def count_rs(*a, **k):
        calls["n"] += 1
        return orig_rs(*a, **k)
    dist.all_reduce, dist.reduce_scatter_tensor = count_ar, count_rs
    out_a = red.reduce_flat(views)
    n_a = calls["n"]
    # (b) unrelated tensors (and a None): one concatenated copy
    calls["n"] = 0
    loose = [torch.randn(7, 3, generator=g), None, torch.randn(5, generator=g), torch.randn(2, 2, 2, generator=g)]
    out_b = red.reduce_flat(loose)
    n_b = calls["n"]
    # (c) only part of an arena's summed entries, with a foreign entry in between: no in-place sum, one copy; (d) SH bands
    calls["n"] = 0
    ar2 = gs2m_arena.GradArena("cpu", [("a", (6, 3)), ("mid", (6, 2)), ("b", (6, 4)), ("sh", (6, 16, 3))], zero=True)
    for name in ("a", "mid", "b", "sh"):
        ar2[name].copy_(torch.randn(ar2[name].shape, generator=g))
    ar2["sh"][:, 4:] = 0.0  # degree 1: bands above it have no gradient on any rank
    mid = ar2["mid"].clone()
    red_sh = GradReducer(mode=mode, sh_active_coeffs=None)
    out_c = red_sh.reduce_flat([ar2["a"], ar2["b"], ar2["sh"]], sh_active={2: 4})
    plan_c = list(red_sh.last_plan)
    untouched = torch.equal(ar2["mid"], mid) and torch.equal(ar["x"], keep_x) and torch.equal(ar["y"], keep_y)
    dist.all_reduce, dist.reduce_scatter_tensor = orig_ar, orig_rs
    q.put((rank, [t.clone().numpy() for t in out_a], [None if t is None else t.clone().numpy() for t in out_b], n_a, n_b,
           all(o is v for o, v in zip(out_a, views)), [t.clone().numpy() for t in out_c], plan_c, untouched))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_one_collective_per_step(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (0 if mode == "allreduce" else 1)
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    res = []
    for _ in range(300):
        try:
            res.append(q.get(timeout=1.0))
        except _queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        if len(res) == 2:
            break
    assert len(res) == 2
    res.sort(key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_a, exp_b, exp_c = None, None, None
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        a = [torch.randn(s, generator=g) for s in ((10, 3), (10, 4), (10, 16, 3))]
        torch.randn((10, 4), generator=g); torch.randn((10, 6), generator=g)  # the two entries the caller keeps to itself
        b = [torch.randn(7, 3, generator=g), None, torch.randn(5, generator=g), torch.randn(2, 2, 2, generator=g)]
        c = [torch.randn(s, generator=g) for s in ((6, 3), (6, 2), (6, 4), (6, 16, 3))]
        c[3][:, 4:] = 0.0
        c = [c[0], c[2], c[3]]
        exp_a = a if exp_a is None else [x + y for x, y in zip(exp_a, a)]
        exp_b = b if exp_b is None else [None if x is None else x + y for x, y in zip(exp_b, b)]
        exp_c = c if exp_c is None else [x + y for x, y in zip(exp_c, c)]
    for rank, out_a, out_b, n_a, n_b, in_place, out_c, plan_c, untouched in res:
        assert n_a == 1 and n_b == 1, "one collective each"
        assert in_place, "arena tensors are reduced in place"
        assert untouched, "entries of an arena that were not passed must not be modified"
        # (c): `a` and `b` have the foreign entry `mid` between them -> one concatenated copy; the SH tensor travels as its
        # 4 active coefficients
        assert sorted(k for k, _ in plan_c) == ["copy", "sh"] and dict(plan_c)["sh"] == 6 * 4 * 3, plan_c
        for got, want in zip(out_a, exp_a):
            assert np.allclose(got, want.numpy(), rtol=1e-6, atol=1e-6)
        for got, want in zip(out_b, exp_b):
            assert (got is None) == (want is None) and (got is None or np.allclose(got, want.numpy(), rtol=1e-6, atol=1e-6))
        for got, want in zip(out_c, exp_c):
            assert np.allclose(got, want.numpy(), rtol=1e-6, atol=1e-6)
