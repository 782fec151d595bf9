#!/usr/bin/env python3
"""bench.py -- train views/s (fwd+bwd raster) at 1M Gaussians 1080p on N MI355X.

One "step" = one view per GPU: GaussianRasterizer forward + backward with dense upstream
gradients on colour and the G-buffer, timed at the op boundary; at N > 1
every rank renders its own camera of the same replicated 1M-Gaussian scene and the step ends
with the RCCL sum of the per-Gaussian gradients (gs2m_dp).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(P, W, H, fc, seed):
    """The CPU oracle (C/OpenMP restatement, kind "port") timed on this box's host cores on a
    bounded sample of the same workload.  Checker infrastructure, never on the product path."""
    import gs2m_synth as S
    from oracle import oracle
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    Ps = P if cores >= 48 else min(P, 250_000)
    cam = S.make_camera(W, H)
    g = S.make_gaussians(P, cam, seed=seed)
    g = {k: v[:Ps].contiguous() for k, v in g.items()}
    Gc, Gb = S.make_upstream_grads(H, W, seed=seed)
    t0 = time.time()
    f = oracle.forward(g["means3D"].numpy(), g["opacities"].numpy(), shs=g["shs"].numpy(), scales=g["scales"].numpy(),
                       rotations=g["rotations"].numpy(), features=g["features"].numpy(),
                       bg=torch.zeros(3).numpy(), viewmatrix=cam["viewmatrix"].numpy(),
                       projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(), W=W, H=H,
                       tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3, feature_count=fc)
    oracle.backward(f, Gc.numpy(), Gb.numpy())
    dt = time.time() - t0
    scale = Ps / P
    return {"value": round(scale / dt, 5), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"1 view fwd+bwd of the first {Ps} of {P} Gaussians at {W}x{H}, fc={fc} "
                      f"({dt:.2f} s on {cores} OpenMP threads" + ("" if Ps == P else f"; value scaled by {Ps}/{P}") + ")"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--fc", type=int, default=9, help="feature_count (9 = --material stage)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-mode", default="allreduce", choices=["allreduce", "rs_ag"])
    ap.add_argument("--bwd-impl", type=int, default=None, choices=[0, 1],
                    help="backward blend implementation (default: the library's default)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} must be launched with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the rasterizer has no CPU path)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("GS2M_DIST_BACKEND", "nccl")  # "gloo": functional check of the N > 1 path on one GPU
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} HIP devices visible")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import gs2m_native
    import gs2m_synth as S
    from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_gaussians
    from gs2m_dp import GradReducer

    if a.bwd_impl is not None:
        gs2m_native.set_bwd_impl(a.bwd_impl)
    P, W, H, fc = a.gaussians, a.width, a.height, a.fc
    if world == 1:
        cam = S.make_camera(W, H)
    else:  # same cloud, cameras on a small arc around its centre: statistically equal per-rank work
        import math
        th = math.radians(3.0 * (rank - (world - 1) / 2.0))
        eye = (6.0 * math.sin(th), 0.0, 6.0 - 6.0 * math.cos(th))
        cam = S.look_at_camera(W, H, eye, (0.0, 0.0, 6.0))
    ref_cam = S.make_camera(W, H)
    g = S.make_gaussians(P, ref_cam, seed=a.seed)
    Gc, Gb = S.make_upstream_grads(H, W, seed=a.seed)
    Gc, Gb = Gc.to(dev), Gb.to(dev)
    prm = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    means2D = torch.zeros(P, 4, device=dev, requires_grad=True)
    st = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
        scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(dev), projmatrix=cam["projmatrix"].to(dev), sh_degree=3,
        campos=cam["campos"].to(dev), prefiltered=False, feature_count=fc)
    empty = torch.Tensor([])
    leaves = [prm["means3D"], means2D, prm["shs"], prm["opacities"], prm["scales"], prm["rotations"], prm["features"]]
    reducer = GradReducer(mode=a.dp_mode)
    if os.environ.get("GS2M_SPIN_WAIT") is not None:  # debugging aid: 0 = hipStreamSynchronize instead of polling the pinned count
        gs2m_native.set_spin_wait(int(os.environ["GS2M_SPIN_WAIT"]))
    info = {}

    def step():
        for t in leaves:
            t.grad = None
        color, radii, observe, buffer = rasterize_gaussians(
            prm["means3D"], means2D, prm["shs"], empty, prm["opacities"], prm["scales"], prm["rotations"], empty,
            prm["features"], st)
        torch.autograd.backward([color, buffer], [Gc, Gb])
        if world > 1:
            # The RCCL sum of this view's gradients (276 MB per rank at M = 16) is started here and waited for at the
            # END of the next step, right before that step starts its own: it runs on RCCL's stream beside the next
            # view's rasterization instead of in front of it.  Every step's gradients are fully reduced (fence()
            # drains the last one inside the timed region); a training loop consumes them one step late.
            drain()
            pending.append(reducer.reduce_grads_async({"means3D": prm["means3D"].grad, "shs": prm["shs"].grad,
                                                       "opacities": prm["opacities"].grad, "scales": prm["scales"].grad,
                                                       "rotations": prm["rotations"].grad, "features": prm["features"].grad}))
            pending.append(reducer.reduce_densification_stats_async(means2D.grad, radii, observe))
        info["radii"] = radii
        info["R"] = color.grad_fn.num_rendered if hasattr(color.grad_fn, "num_rendered") else None

    pending = []

    def drain():
        while pending:
            pending.pop(0).wait()

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    gs2m_native.profile_mode(1)  # HIP events around the two blend kernels, on their launch stream
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    t1 = time.perf_counter()
    blend = gs2m_native.profile_collect()
    ms = (t1 - t0) / a.steps * 1e3
    if world > 1:
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())

    # untimed: per-stage breakdown of the same step
    gs2m_native.profile_mode(2)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    stages = gs2m_native.profile_collect()
    gs2m_native.profile_mode(0)

    if rank == 0:
        V = int((info["radii"] > 0).sum().item())
        R = int(info["R"]) if info["R"] is not None else -1
        Tn = ((W + 15) // 16) * ((H + 15) // 16)
        ab = S.algo_bytes(P, V, R, W * H, Tn, fc)
        k_ms = {k: blend[k][0] / max(blend[k][1], 1) for k in ("blend_fwd", "blend_bwd")}
        dom = max(k_ms, key=k_ms.get)
        achieved = ab[dom] / (k_ms[dom] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # filled from rocprofv3 --pmc passes
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "train views/s (fwd+bwd raster) at 1M Gaussians 1080p",
            "value": round(world * 1e3 / ms, 3), "unit": "views/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{P} synthetic Gaussians (SH deg 3), 1 camera {W}x{H} per GPU, feature_count={fc}, "
                                   f"fwd+bwd at the op boundary" + ("" if world == 1 else ", RCCL gradient sum per step (overlapping the next view)"),
                       "gaussians": P, "visible": V, "num_rendered": R, "width": W, "height": H, "feature_count": fc,
                       "parallelism": f"view-parallel x{world}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algo_bytes_per_launch": ab[dom], "avg_launch_ms": round(k_ms[dom], 5),
                         "blend_fwd_ms": round(k_ms["blend_fwd"], 5), "blend_bwd_ms": round(k_ms["blend_bwd"], 5),
                         "whole_path_GBps": round(ab["total"] / (ms * 1e-3) / 1e9, 2)},
            "stages_ms": {k: round(v[0] / max(v[1], 1), 5) for k, v in stages.items()},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, W, H, fc, a.seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
