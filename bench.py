#!/usr/bin/env python3
"""bench.py -- train views/s (fwd+bwd raster) at 1M Gaussians 1080p on N MI355X.

One "step" = one view per GPU: GaussianRasterizer forward + backward with dense upstream gradients on colour and
the G-buffer, timed at the op boundary.  At N > 1 (north_star / BASELINE configs[4]) every rank renders ONE
camera of the same replicated scene -- rank r: position r of SURVEY.md 8(d)'s 8-position ring -- and the step ENDS with the
blocking RCCL sum of the view's per-Gaussian gradients -- one collective over the binding's gradient arena -- plus the
densification side channels (gs2m_dp): that is `value`.  Beside it, timed in the same run and labelled: the same step on
equal-work cameras (`equal_work_*`), with two views per rank whose gradients accumulate before ONE reduction
(`accumulate_v2_*`), with the reduction of step k overlapping step k + 1 (`pipelined_*`), and without any collective
(`compute_only_*`: what N independent GPUs would deliver -- the reference point for scaling efficiency on THIS workload).
Prints ONE JSON line on rank 0.

Order inside a run: W warm-up steps, K steps timed "at start" (`clock_ramp`: the GPU's clock governor is still ramping
there), the auxiliary passes of the same step (pipelined / one-view forms at N > 1, the reference-binning pass), then W warm-up
steps again and the K timed steps `value` is computed from, then the per-stage pass (`stages_ms`).

  --config c3            BASELINE.json configs[2]: 1M Gaussians, 1920x1080, feature_count 9 (material G-buffers) -- the
                         configuration the metric is quoted on; the default at N = 1
  --config c2            configs[1]: 500k Gaussians, 1080p, feature_count 5 (colour + depth + normal)
  --config c5            configs[4]: 2M Gaussians, 1080p, feature_count 9, one view per GPU -- the default at N > 1 (the data-
                         parallel configuration BASELINE names; `--config c3` keeps the single-GPU workload at every N)
  --config c1            configs[0]: 10k Gaussians, 256x256, feature_count 10 (the CPU-runnable case)
  --config c4            configs[3]: the full train.py loop at DTU scan24's size on a SUBSTITUTE scene (the dataset is not on the
                         box): 49 views 777x581 through the COLMAP loader at -r 2, ~30 k -> past 300 k points, geometry stage with
                         lambda_depth_normal 0.015 and the multi-view term; prints training iterations/s (N = 1 only)
At N = 1 the line also carries `cpu_baseline` (the oracle timed on this box's host cores), `render_level_ms`,
`train_step_ms` and `material_step_ms` (render(), a full geometry-stage and a full material-stage training iteration around
the same op).
"""
import argparse
import glob
import hashlib
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
SIMDS = 1024           # 256 CUs x 4 SIMDs
CONFIGS = {  # name -> (Gaussians, width, height, feature_count)
    "c1": (10_000, 256, 256, 10), "c2": (500_000, 1920, 1080, 5), "c3": (1_000_000, 1920, 1080, 9),
    "c5": (2_000_000, 1920, 1080, 9)}


def kernel_source_hash():
    """sha256 over the DEVICE code of the library the bench runs on -- the `.hip_fatbin` section of libgs2m_raster.so (every kernel's
    code object): ties a counter file to the kernels it was taken on.  A host-side edit (api.hip's launch logic, a comment) leaves it
    unchanged, any change of a kernel or of its compile flags changes it.  (Rounds 1-5 hashed the source FILES, so a host-only edit
    invalidated the counters and the stamp had to be renewed by hand.)"""
    import struct
    path = os.environ.get("GS2M_LIB", os.path.join(ROOT, "gs-2m_amd", "csrc", "libgs2m_raster.so"))
    try:
        blob = open(path, "rb").read()
        assert blob[:4] == b"\x7fELF" and blob[4] == 2  # ELF64
        shoff, = struct.unpack_from("<Q", blob, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
        sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)
        str_off = sec(shstrndx)[4]
        h = hashlib.sha256()
        found = False
        for i in range(shnum):
            name_off, _, _, _, off, size = sec(i)[:6]
            name = blob[str_off + name_off: blob.index(b"\0", str_off + name_off)].decode()
            if name == ".hip_fatbin":
                h.update(blob[off: off + size])
                found = True
        assert found
        return h.hexdigest()[:16]
    except Exception:
        return "no-device-code"


def counters_for(kernel, workload):
    """PMC counters of `kernel` ("blend_fwd" / "blend_bwd") from profiles/pmc_counters.json -- written by
    tools/profile.sh + tools/summarize_prof.py from separate `rocprofv3 --pmc` passes over this very command -- or
    None when the file was taken on other kernel sources or another workload (a stale file is refused, not trusted)."""
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    if d.get("source_hash") != kernel_source_hash() or d.get("workload") != workload:
        return None
    return d.get("kernels", {}).get(kernel)


def cpu_baseline(P, W, H, fc, seed):
    """The CPU oracle (C/OpenMP restatement, kind "port") timed on this box's host cores on a
    bounded sample of the same workload.  Checker infrastructure, never on the product path."""
    import gs2m_synth as S
    from oracle import oracle
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    flags = oracle.use_native_build() or "-O2 -fopenmp (native build not possible here)"  # -O3 -march=native, built on this box
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
    out = {"value": round(scale / dt, 5), "unit": "views/s", "cores": cores, "kind": "port",
           "sample": f"1 view fwd+bwd of the first {Ps} of {P} Gaussians at {W}x{H}, fc={fc} "
                     f"({dt:.2f} s on {cores} OpenMP threads, oracle built here with {flags}" + ("" if Ps == P else f"; value scaled by {Ps}/{P}") + ")"}
    if P <= 20_000:
        # also the PyTorch-autograd variant (tests/torch_ref.py, float64, single process) -- the CPU
        # autograd rasterizer BASELINE.json configs[0] names
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from torch_ref import rasterize_dense
        torch.set_num_threads(min(8, cores))  # hundreds of threads on ops this small only contend (13 minutes instead of half of one)
        d = {k: v.double().requires_grad_(True) for k, v in g.items()}
        t0 = time.time()
        color, buf, _, _ = rasterize_dense(d["means3D"], d["opacities"], d["shs"], None, d["scales"], d["rotations"], None, d["features"],
                                           bg=torch.zeros(3), viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"],
                                           W=W, H=H, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3, feature_count=fc)
        ((color * Gc.double()).sum() + (buf * Gb.double()).sum()).backward()
        dta = time.time() - t0
        out["autograd"] = {"value": round(1.0 / dta, 5), "unit": "views/s", "kind": "PyTorch CPU autograd restatement (tests/torch_ref.py), float64, "
                           f"{torch.get_num_threads()} torch threads", "seconds": round(dta, 2)}
    return out


def caller_levels(P, W, H, seed, dev, steps=10, warmup=6):
    """the same op one and two levels up.  render_level_ms: gaussian_renderer.render(material stage)
    forward + backward of a weighted sum of its maps.  train_step_ms: one geometry-stage training iteration
    (train.py:94-130, 223-227, 258-259 without the multi-view term): render with the Sobel normal, clamp,
    L1 + D-SSIM + plane + depth-normal losses, backward, densification statistics, fused Adam step.
    material_step_ms: one material-stage iteration (train.py:132-196 without the multi-view roughness term, config C3's
    "+ deferred pbr.shade"): render, prefilter of a learnable 512^2 environment light, deferred split-sum shading, L1 + D-SSIM
    on the shaded image, depth-normal, plane and the three edge-aware smoothness terms, backward to Gaussians and light,
    statistics, both Adam steps."""
    import gs2m_optim
    import gs2m_synth as S
    from fused_ssim import dssim_loss
    from gaussian_renderer import render
    from gs2m_losses import densification_stats, edge_gradient, fused_plane_loss, fused_tv_loss, geometry_image_loss
    from gs2m_scene import Camera, GaussianParams, PipelineParams
    from pbr import CubemapLight, get_brdf_lut, pbr_render
    cam0 = S.make_camera(W, H)
    g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam0, seed=seed).items()}
    u = lambda c: torch.rand(P, c, device=dev) * 0.8 + 0.1
    pc = GaussianParams.from_activated(g["means3D"], g["shs"], g["scales"], g["rotations"], g["opacities"].clamp(0.01, 0.99), u(3), u(1), u(1))
    params = [torch.nn.Parameter(t) for t in pc.parameters()]
    (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity, pc._albedo, pc._roughness, pc._metallic) = params
    names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic")
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3, albedo=0.05, roughness=0.05, metallic=0.05)
    opt = gs2m_optim.Adam([{"params": [p], "lr": lrs[n], "name": n} for p, n in zip(params, names)], lr=0.0, eps=1e-15)
    cam, pipe, bg = Camera(cam0, dev), PipelineParams(), torch.zeros(3, device=dev)
    wts = {k: torch.rand(s, device=dev) for k, s in (("render", (3, H, W)), ("depth_map", (1, H, W)), ("normal_map", (3, H, W)),
                                                     ("albedo_map", (3, H, W)), ("roughness_map", (1, H, W)), ("local_normal_map", (3, H, W)))}
    gt = torch.rand(3, H, W, device=dev)
    accum, accum_abs, denom = (torch.zeros(P, 1, device=dev) for _ in range(3))
    state = {"max_radii": torch.zeros(P, device=dev)}

    def render_step():
        for t in params:
            t.grad = None
        out = render(cam, pc, pipe, bg, material_stage=True)
        sum((out[k] * w).sum() for k, w in wts.items()).backward()

    def train_step():
        out = render(cam, pc, pipe, bg, geometry_stage=False, material_stage=False, sobel_normal=True)
        vis, radii = out["visibility_filter"], out["radii"]
        # clamp + L1 + edge-weighted depth-normal term in one pass (the edge strength of the ground truth recomputed per
        # iteration, as the reference does), plane term and densification statistics as one launch each (csrc/loss_ops.hip)
        rgb, Limg, _ = geometry_image_loss(out["render"], gt, out["normal_map"], out["sobel_map"], edge=edge_gradient(gt), w_l1=0.8, w_dn=0.015)
        loss = Limg + dssim_loss(rgb.unsqueeze(0), gt.unsqueeze(0), 0.2) + fused_plane_loss(vis, pc, weight=0.01)
        loss.backward()
        with torch.no_grad():  # train.py:223-227, GM:569-573
            densification_stats(out["viewspace_points"].grad, vis, accum, accum_abs, denom, out["observe"], radii, state["max_radii"])
            opt.step()
            opt.zero_grad(set_to_none=True)

    class Lighting:  # what pbr_render needs of the reference's Scene (scene/__init__.py:44-46, 144-148)
        cubemap = CubemapLight(base_res=512, device=dev)
        brdf_lut = get_brdf_lut().to(dev)
    light_opt = gs2m_optim.Adam([{"name": "cubemap", "params": list(Lighting.cubemap.parameters()), "lr": 0.05}], lr=0.05)
    rays = torch.nn.functional.normalize(cam.get_rays().view(-1, 3), p=2, dim=-1)
    edge = edge_gradient(gt)  # per view, as gs2m_train keeps it

    def material_step():
        out = render(cam, pc, pipe, bg, geometry_stage=True, material_stage=True, sobel_normal=True)
        vis, radii = out["visibility_filter"], out["radii"]
        pkg = pbr_render(Lighting, cam, rays, out, metallic=False)
        pbr, Limg, _ = geometry_image_loss(pkg["render_rgb"], gt, out["normal_map"], out["sobel_map"], edge=edge, w_l1=0.8, w_dn=0.015,
                                           mask=out["normal_mask"], background=bg)
        wn = (0.5 * torch.tanh(8.0 * ((1.0 - out["roughness_map"]).detach() - 0.5)) + 0.5).clamp(0, 1)
        loss = (Limg + dssim_loss(pbr.unsqueeze(0), gt.unsqueeze(0), 0.2) + fused_plane_loss(vis, pc, weight=0.01)
                + fused_tv_loss(gt, out["roughness_map"], norm1=False, weight=0.002) + fused_tv_loss(gt, out["albedo_map"], weight=0.01)
                + fused_tv_loss(gt, out["normal_map"], weight_map=wn, weight=0.01))
        loss.backward()
        with torch.no_grad():
            densification_stats(out["viewspace_points"].grad, vis, accum, accum_abs, denom, out["observe"], radii, state["max_radii"])
            opt.step()
            opt.zero_grad(set_to_none=True)
            light_opt.step()
            light_opt.zero_grad(set_to_none=True)
            Lighting.cubemap.clamp_(min=0.0)

    res = {}
    start = [t.detach().clone() for t in params]
    for key, fn in (("render_level_ms", render_step), ("train_step_ms", train_step), ("material_step_ms", material_step)):
        with torch.no_grad():  # every level starts from the same scene: the Adam steps of the previous one have moved it
            for t, t0 in zip(params, start):
                t.copy_(t0)
        for _ in range(warmup):
            fn()
        # three blocks of `steps` iterations, the fastest block reported: one allocator growth (a hipMalloc of the caching
        # allocator: milliseconds) inside a block of 20 moved this number by 25 % from box to box
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            best = ms if best is None else min(best, ms)
        res[key] = round(best, 4)
    return res


def bench_c4(a):
    """BASELINE.json configs[3] on the substitute scene (gs2m_train.c4_scene / c4_run; tests/test_c4_gpu.py checks the same run):
    value = training iterations per second over the whole run (render, losses, backward, densify / prune / reset / trim, Adam),
    the rasterizer's share of an iteration measured on the final model, and the CPU oracle's fwd+bwd of one training view of
    the final model beside it."""
    import tempfile
    import numpy as np
    import gs2m_native
    import gs2m_synth as S
    import gs2m_train
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    iters = a.c4_iterations
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        scene = gs2m_train.c4_scene(os.path.join(tmp, "c4"), seed=a.seed)
        t_scene = time.perf_counter() - t0
        for _ in range(max(1, a.warmup // 25)):  # warm-up: a short run (allocator, clocks, lazily loaded kernels)
            gs2m_train.c4_run(None, iterations=150, schedule_iterations=iters, scene=scene, seed=a.seed)
        model, st = gs2m_train.c4_run(None, iterations=iters, scene=scene, seed=a.seed)
    cams = scene[0]
    W, H = cams[0].image_width, cams[0].image_height
    # the rasterizer inside one iteration of the final model: stage times of view 0's render + backward
    from gaussian_renderer import render
    from gs2m_scene import PipelineParams
    pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
    gs2m_native.profile_mode(2)
    info = {}
    for _ in range(5):
        for p in model.parameters():
            p.grad = None
        out = render(cams[0], model, pipe, bg, True, False, sobel_normal=True)
        (out["render"].sum() + out["depth_map"].sum() + out["normal_map"].sum()).backward()
        info["radii"] = out["radii"]
    torch.cuda.synchronize()
    stages = gs2m_native.profile_collect()
    gs2m_native.profile_mode(0)
    P = int(model.get_xyz.shape[0])
    V = int((info["radii"] > 0).sum().item())
    Tn = ((W + 15) // 16) * ((H + 15) // 16)
    k_ms = {k: v[0] / max(v[1], 1) for k, v in stages.items()}
    raster_ms = sum(v[0] for v in stages.values()) / 5.0
    fc = 5  # geometry stage: colour + alpha + distance + normal
    R = None
    out_line = {
        "metric": "train iterations/s, full train.py loop (BASELINE configs[3]; SUBSTITUTE scene: DTU scan24 is not on the box)",
        "value": round(st["it_per_s"], 3), "unit": "iterations/s", "n_gpus": 1, "steps": iters, "warmup": a.warmup,
        "ms_per_step": round(1e3 / st["it_per_s"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": st["workload"], "views": len(cams), "width": W, "height": H, "points_start": st["points_start"],
                   "points_max": st["points_max"], "points_end": st["points_end"], "psnr_start": round(st["psnr_start"], 3),
                   "psnr_end": round(st["psnr_end"], 3), "scene_build_s": round(t_scene, 2)},
        "gpu_busy_frac": None,
        "rasterizer_ms_per_iteration_at_end": round(raster_ms, 4),
        "stages_ms_at_end": {k: round(v, 5) for k, v in k_ms.items() if v},
    }
    # GPU-busy fraction: the kernel time of exactly this run (the trajectory is deterministic) from a kernel trace of the same kernels
    # (tools/c4_busy.sh -> profiles/c4_kernel_time.json, refused when taken on other device code) over this run's wall time
    try:
        kt = json.load(open(os.path.join(ROOT, "profiles", "c4_kernel_time.json")))
        if kt.get("source_hash") == kernel_source_hash() and kt.get("iterations") == iters:
            out_line["gpu_busy_frac"] = round(kt["kernel_ms_total"] / (1e3 * iters / st["it_per_s"]), 4)
            out_line["gpu_busy_source"] = (f"sum of the kernel durations of the same {iters} iterations under rocprofv3 --kernel-trace ({kt['kernel_us_per_iteration']} us and "
                                           f"{kt['launches_per_iteration']} launches per iteration; profiles/c4_kernel_time.json) / this run's untraced wall time")
    except Exception:
        pass
    dom = "blend_bwd"
    if not a.no_cpu_baseline:
        # the CPU oracle on what render() hands the op for view 0 of the final model (checker infrastructure, never on the product path)
        from test_c4_gpu import _capture_rasterizer_call
        import helpers as Hh
        from oracle import oracle
        cores = os.cpu_count() or 1
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        flags = oracle.use_native_build() or "-O2 -fopenmp"
        sc = _capture_rasterizer_call(cams[0], model, geometry_stage=True)
        t0 = time.time()
        f, gr = Hh.run_oracle(oracle, sc)
        dt = time.time() - t0
        R = int(f.num_rendered)
        out_line["cpu_baseline"] = {"value": round(1.0 / dt, 4), "unit": "iterations/s (rasterizer forward + backward of ONE training view only: an upper bound for a CPU loop)",
                                    "cores": cores, "kind": "port",
                                    "sample": f"oracle fwd+bwd of view 0 of the final model ({P} points, {W}x{H}, fc {sc['fc']}; {dt:.2f} s on {cores} OpenMP threads, built with {flags})"}
    if R is not None:
        ab = S.algo_bytes(P, V, R, W * H, Tn, fc)
        out_line["config"]["num_rendered_view0_reference_binning"] = R
        ach = ab[dom] / (k_ms[dom] * 1e-3) / 1e9
        out_line["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                                "traffic": None, "algo_bytes_per_launch": ab[dom], "avg_launch_ms": round(k_ms[dom], 5),
                                "measured": "stage pass on view 0 of the final model (algorithmic bytes priced with the reference's instance count)"}
    print(json.dumps(out_line), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` typed as such (no launcher, WORLD_SIZE unset): start the N ranks as a CHILD process --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>` --
    before this process has made any GPU call (importing torch does not initialise HIP), relay what the ranks print (rank 0's
    JSON line among it) and return the child's exit code.  No exec: the parent stays a plain waiting process."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between the ranks' processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS) + ["c4"], help="BASELINE.json configuration (see the module docstring); default c3 on one GPU, c5 on several")
    ap.add_argument("--c4-iterations", type=int, default=5000, help="--config c4: training iterations (the reference runs 30000; the schedule is compressed)")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--fc", type=int, default=None, help="feature_count (9 = --material stage)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-caller-levels", action="store_true", help="skip render_level_ms / train_step_ms / material_step_ms")
    ap.add_argument("--dp-mode", default="auto", choices=["auto", "allreduce", "rs_ag"],
                    help="auto: reduce-scatter + all-gather from 4 ranks on (every link of the xGMI mesh busy), all-reduce below")
    ap.add_argument("--equal-work", action="store_true",
                    help="N > 1: `value` on the equal-work arc (the single-GPU camera moved 0.4 degrees per view around the cloud centre: per-GPU "
                         "work stays what it is at N = 1) instead of the default, the 8-position camera ring of SURVEY.md 8(d) (45 degrees apart: "
                         "the views differ in work by up to 1.6x); the other one is timed beside it either way")
    ap.add_argument("--ring-position", type=int, default=None,
                    help="N = 1 only: render the camera rank k of an N-GPU run takes (position k of the 8-camera ring, SURVEY.md 8(d)) "
                         "-- the per-view times the multi-GPU model in DESIGN.md section 6 is built from; not the metric's workload")
    ap.add_argument("--split-sh", action="store_true", help="experiment: hand the SH coefficients over as the model stores them, DC (P,1,3) "
                    "and rest (P,15,3) (this repository's shs_rest extension), instead of the reference op's one (P,16,3) tensor")
    ap.add_argument("--heavy-tail", default=None, metavar="F:K",
                    help="not the metric's workload: a fraction F of the Gaussians K times larger (splats over hundreds of tiles, as "
                         "close-ups and background blobs of real scenes have them) -- how the stages hold up off the uniform scene")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="views every rank renders per step of the HEADLINE (default 1: the metric's step, north_star's form); with more "
                         "than one their gradients ACCUMULATE and one reduction follows the last view (exact sums, no stale gradients; the "
                         "collective is paid once per that many views).  The two-view form is timed beside the headline at N > 1 anyway")
    ap.add_argument("--no-reference-binning", action="store_true", help="skip the second timing of the same workload in reference-binning mode")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.config is None:
        a.config = "c3" if world == 1 else "c5"
    # watchdog: a rank that hangs (a peer died inside a collective, a stuck kernel) dumps every thread's stack and exits
    # instead of holding the GPU until the caller's limit
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("GS2M_BENCH_WATCHDOG_S", "900")), exit=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} must be launched with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the rasterizer has no CPU path)")
    if a.config == "c4":
        if world != 1:
            raise SystemExit("bench.py --config c4 is a single-GPU run")
        return bench_c4(a)
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

    P, W, H, fc = CONFIGS[a.config]
    P, W, H, fc = a.gaussians or P, a.width or W, a.height or H, fc if a.fc is None else a.fc
    preset = (P, W, H, fc) == CONFIGS[a.config]
    # the N-GPU workload renders cameras on a circle of radius 6 around the cloud centre (0, 0, 6), looking at
    # it, 8 positions 45 degrees apart; rank r takes position r (position 0 is the single-GPU camera at the origin).  The
    # views differ in work (instances per view): the step time is the slowest rank's.
    def ring_camera(pos):
        if pos % 8 == 0:
            return S.make_camera(W, H)
        import math
        th = 2.0 * math.pi * (pos % 8) / 8.0
        eye = (6.0 * math.sin(th), 0.0, 6.0 - 6.0 * math.cos(th))
        return S.look_at_camera(W, H, eye, (0.0, 0.0, 6.0))
    def arc_camera(k):  # the single-GPU camera moved k x 0.4 degrees on the circle around the cloud centre: the same work per view
        if k == 0:
            return S.make_camera(W, H)
        import math
        th = math.radians(0.4 * k)
        return S.look_at_camera(W, H, (6.0 * math.sin(th), 0.0, 6.0 - 6.0 * math.cos(th)), (0.0, 0.0, 6.0))
    Vn = max(1, a.views_per_rank)
    # rank r, view v of a step: camera r + v * world of the ring (default at N > 1) or of the arc; world == 1: the single-GPU camera
    # first.  The headline's cameras and, at N > 1, the other family beside it; two views per rank for the accumulate form.
    head_ring = (world > 1 and not a.equal_work) or (world == 1 and a.ring_position is not None)
    first = rank if world > 1 else (a.ring_position or 0)
    nviews = max(Vn, 2) if world > 1 else Vn
    fam = {"ring": [ring_camera(first + v * world) for v in range(nviews)] if (world > 1 or head_ring) else None,
           "arc": [arc_camera(first + v * world) for v in range(nviews)] if (world > 1 or not head_ring) else None}
    head = "ring" if head_ring else "arc"
    cam = fam[head][0]
    ref_cam = S.make_camera(W, H)
    g = S.make_gaussians(P, ref_cam, seed=a.seed)
    if a.heavy_tail:
        frac, factor = (float(x) for x in a.heavy_tail.split(":"))
        big = torch.rand(P, generator=torch.Generator().manual_seed(a.seed + 1)) < frac
        g["scales"] = torch.where(big[:, None], g["scales"] * factor, g["scales"])
        preset = False
    Gc, Gb = S.make_upstream_grads(H, W, seed=a.seed)
    Gc, Gb = Gc.to(dev), Gb.to(dev)
    prm = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    means2D = torch.zeros(P, 4, device=dev, requires_grad=True)
    mk = lambda c: GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=torch.zeros(3, device=dev),
        scale_modifier=1.0, viewmatrix=c["viewmatrix"].to(dev), projmatrix=c["projmatrix"].to(dev), sh_degree=3,
        campos=c["campos"].to(dev), prefiltered=False, feature_count=fc)
    sts_f = {k: [mk(c) for c in v] for k, v in fam.items() if v is not None}
    empty = torch.Tensor([])
    sh_in, sh_rest = prm["shs"], None
    if a.split_sh:
        sh_in = prm["shs"][:, :1].detach().clone().requires_grad_(True)
        sh_rest = prm["shs"][:, 1:].detach().clone().requires_grad_(True)
    leaves = [prm["means3D"], means2D, sh_in, prm["opacities"], prm["scales"], prm["rotations"], prm["features"]]
    if sh_rest is not None:
        leaves.append(sh_rest)
    dp_mode = ("rs_ag" if world >= 4 else "allreduce") if a.dp_mode == "auto" else a.dp_mode
    reducer = GradReducer(mode=dp_mode)
    VPR = Vn  # views per rank and step of the headline
    import gs2m_arena
    gs2m_arena.set_keep(max(VPR, 2) + 1)  # accumulate mode: the first view's arena (the accumulated gradients) stays registered until the reduction
    if world > 1:
        gs2m_native.set_sort_tickets(True)  # RCCL's kernels share the device: the tile sort must not assume it has the GPU to itself
    if os.environ.get("GS2M_SPIN_WAIT") is not None:  # debugging aid: 0 = hipStreamSynchronize instead of polling the pinned count
        gs2m_native.set_spin_wait(int(os.environ["GS2M_SPIN_WAIT"]))
    info = {}
    pending = []

    def drain():
        while pending:
            pending.pop(0).wait()

    def step(pipelined=False, family=None, views=None, reduce=True):
        sts = sts_f[family or head]
        nv = views or VPR
        for t in leaves:
            t.grad = None
        local = None
        for v in range(nv):
            if v:  # dL/dmeans2D is per view (densification accumulates NORMS of it, train.py:223-227): not accumulated
                means2D.grad = None
            color, radii, observe, buffer = rasterize_gaussians(
                prm["means3D"], means2D, sh_in, empty, prm["opacities"], prm["scales"], prm["rotations"], empty,
                prm["features"], sts[v], sh_rest)
            torch.autograd.backward([color, buffer], [Gc, Gb])  # v > 0: autograd ADDS to the first view's gradients, in its arena
            if world > 1 and nv > 1 and reduce:
                with torch.no_grad():
                    local = reducer.local_densification_stats(means2D.grad, radii, observe, into=local)
        if world > 1 and reduce:
            # The sum of this step's gradients: ONE collective over the arena the binding allocated them in (the leaves'
            # .grad are views of it; with several views per rank they hold the views' accumulated sums), plus the
            # densification side channels.  Blocking form (the metric): the step ends when the sums have landed.
            # Pipelined form: started here, waited for right before the NEXT step's own reduction starts, so it runs on
            # RCCL's stream beside the next step's rasterization; every step's gradients are still fully reduced, a
            # training loop built that way applies them one step late.
            if pipelined:
                drain()
            # the densification side channels first: per-view norms of THIS rank's dL/dmeans2D (train.py:223-227), computed
            # before anything is summed; dL/dmeans2D itself is not part of the summed range (its sum is never used)
            pending.append(reducer.reduce_densification_stats_async(means2D.grad, radii, observe, local=local))
            pending.append(reducer.reduce_flat_async([t.grad for t in leaves if t is not means2D]))
            if not pipelined:
                drain()
        info["radii"] = radii
        info["R"] = color.grad_fn.num_rendered if hasattr(color.grad_fn, "num_rendered") else None

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(pipelined=False, **kw):
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(pipelined, **kw)
        fence()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        if world > 1:
            t = torch.tensor([ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        return ms

    # Python's cyclic collector: a full (generation-2) collection walks every tracked object of the process -- 36 ms with the ~170 k
    # objects of `import torch` + this script (tools/stall_probe.py) -- and lands wherever the allocation counters trip, i.e. now and
    # then inside a timed region of 0.25 s.  Everything alive at this point is set-up state that stays alive: moved to the permanent
    # generation, later collections only look at what the loop itself creates (same results; `gs2m_train.train` does the same).
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(a.warmup):
        step()
    # The position rounds 1-3 timed at: right after the warm-up, i.e. inside the clock governor's ramp (sclk takes about half
    # a second of continuous work to reach 2400 MHz, tools/clock_trace.py; the ALU-bound blend kernels run by the clock ratio
    # slower until then).  Kept as `ms_per_step_at_start`; the auxiliary passes below all run the same step, and the headline
    # region is timed after them, again behind W warm-up steps, at the clock a training run of 30k steps spends its time at.
    ms_start = timed(False)
    beside = {}
    if world > 1:  # beside the headline, same run, each behind two untimed steps of its own form
        other = "arc" if head == "ring" else "ring"
        for key, kw in (("pipelined", dict(pipelined=True)), ("equal_work" if other == "arc" else "camera_ring", dict(family=other)),
                        ("accumulate_v2", dict(views=2)), ("compute_only", dict(reduce=False))):
            for _ in range(2):
                step(**kw)
            beside[key] = (timed(**kw), kw.get("views", VPR))
        step()

    # The same workload with the reference's own instance list (gs2m_set_reference_binning(1): every tile of the radius
    # rectangle, auxiliary.h:44-53 -- the mode whose sorted lists are bit-identical to the reference's).  The headline below
    # runs the default mode: a result-identical, order-preserving SUBSET of that list (tiles the alpha >= 1/255 ellipse cannot
    # reach are not emitted).
    ref_binning = None
    if world == 1 and not a.no_reference_binning:
        gs2m_native.lib().gs2m_set_reference_binning(1)
        for _ in range(max(3, a.warmup // 2)):
            step()
        import diff_gaussian_rasterization as _dgr
        c0 = dict(_dgr.BINNING_CACHE_STATS)
        ms_ref = timed(False)
        ref_binning = {"ms_per_step": round(ms_ref, 4), "value": round(VPR * 1e3 / ms_ref, 3),
                       "binning_cache": {k: _dgr.BINNING_CACHE_STATS[k] - c0[k] for k in c0},
                       "num_rendered": int(info["R"]) if info["R"] is not None else -1}
        gs2m_native.lib().gs2m_set_reference_binning(0)

    # The headline: W untimed warm-up steps, then exactly K timed ones.
    # HIP events on the launch stream around the dominant kernel (the backward blend) inside the timed region.  An event pair
    # leaves ~6 us of bubble on the stream on either side of the kernel it brackets (kernel trace: the only gaps of the step), so
    # every BRACKET_EVERY-th launch is bracketed -- `roofline.launches_bracketed` of the K x views launches -- and the other
    # stages are timed in the untimed pass below
    BRACKET_EVERY = 4
    for _ in range(a.warmup):
        step()
    gs2m_native.profile_mode(3, every=BRACKET_EVERY)
    ms = timed(False)
    blend = gs2m_native.profile_collect()
    gs2m_native.profile_mode(0)
    # the same K steps once more with an event between them: per-step times, for the median beside the mean (an event is a
    # marker on the launch stream; the steps run back to back as before, minus ~1 us of marker per step)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    fence()
    evs[0].record()
    for k in range(a.steps):
        step()
        evs[k + 1].record()
    fence()
    per_step = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(a.steps))
    ms_median = per_step[len(per_step) // 2] if a.steps % 2 else 0.5 * (per_step[a.steps // 2 - 1] + per_step[a.steps // 2])

    # untimed: per-stage breakdown of the same step, right behind the headline region (same clock).  Every stage is bracketed by
    # HIP events of its own: a bracket also sees the dispatch latency that back-to-back kernels overlap with their predecessor's
    # tail, so the stage times add up to slightly MORE than ms_per_step (by about 2-4 us per bracket).
    STAGE_STEPS = 5
    gs2m_native.profile_mode(2)
    for _ in range(STAGE_STEPS):
        step()
    fence()
    stages = gs2m_native.profile_collect()
    gs2m_native.profile_mode(0)

    if rank == 0:
        V = int((info["radii"] > 0).sum().item())
        R = int(info["R"]) if info["R"] is not None else -1
        Tn = ((W + 15) // 16) * ((H + 15) // 16)
        ab = S.algo_bytes(P, V, R, W * H, Tn, fc)
        k_ms = {"blend_bwd": blend["blend_bwd"][0] / max(blend["blend_bwd"][1], 1),            # live, timed region
                "blend_fwd": stages["blend_fwd"][0] / max(stages["blend_fwd"][1], 1)}          # untimed stage pass
        # The backward blend dominates at every BASELINE configuration and is the kernel bracketed inside the timed
        # region.  On small frames (a few tiles, e.g. the 320x192 case of tests/test_bench_gpu.py) the forward can be the
        # longer one: it is then reported as the dominant kernel from the stage pass, and `measured` says so.
        dom = "blend_bwd" if k_ms["blend_bwd"] >= k_ms["blend_fwd"] else "blend_fwd"
        measured = "timed region" if dom == "blend_bwd" else "stage pass (untimed, same step)"
        achieved = ab[dom] / (k_ms[dom] * 1e-3) / 1e9
        workload = f"{P}x{W}x{H}x{fc}"
        ctr = counters_for(dom, workload) or {}
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "measured": measured,
                # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command
                # (2 x FETCH + WRITE per the guide's gfx950 correction); null when profiles/pmc_counters.json was taken
                # on other kernel sources or another workload
                "traffic": ctr.get("hbm_bytes"),
                "algo_bytes_per_launch": ab[dom], "avg_launch_ms": round(k_ms[dom], 5),
                "launches_bracketed": int(blend["blend_bwd"][1]) if dom == "blend_bwd" else int(stages["blend_fwd"][1]),
                "blend_fwd_ms": round(k_ms["blend_fwd"], 5), "blend_bwd_ms": round(k_ms["blend_bwd"], 5),
                "whole_path_GBps": round(VPR * ab["total"] / (ms * 1e-3) / 1e9, 2)}
        if ctr.get("SQ_WAVE_CYCLES"):
            # What the dominant kernel's waves spent their time on, from the SQ counters of the same command (separate --pmc
            # passes): shares of SQ_WAVE_CYCLES -- issuing (SQ_ACTIVE_INST_ANY), stalled at issue on a dependency or a busy
            # pipe (SQ_WAIT_INST_ANY), parked in s_waitcnt / barriers (SQ_WAIT_ANY); the three are disjoint.  Rounds 1-3
            # derived an "issue bound" from SQ_INSTS_VALU x 4 cycles: the per-instruction cost measured on this part is
            # 2.6-3.2 cycles (4.2-5.3 packed), the product exceeded the forward's own run time, and the figure is gone.
            wc = float(ctr["SQ_WAVE_CYCLES"])
            roof["counters"] = {k: ctr.get(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS",
                                                        "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE") if ctr.get(k) is not None}
            roof["counters"].update({
                "issuing_share": round(ctr.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
                "issue_stall_share": round(ctr.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                "waitcnt_share": round(ctr.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                "lds_issue_stall_share": round(ctr.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4)})
            if ctr.get("GRBM_GUI_ACTIVE") and ctr.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                # matrix-pipe busy cycles per SIMD over the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                roof["counters"]["mfma_busy_share"] = round(ctr["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / (ctr["GRBM_GUI_ACTIVE"] / 8.0), 4)
        out = {
            # BASELINE.json's metric is quoted on C3 (1M Gaussians, 1080p); other configurations say what they ran
            "metric": ("train views/s (fwd+bwd raster) at 1M Gaussians 1080p" if (P, W, H) == (1_000_000, 1920, 1080) else
                       f"train views/s (fwd+bwd raster) at {P} Gaussians {W}x{H}"),
            "value": round(world * VPR * 1e3 / ms, 3), "unit": "views/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms, 4), "median_ms_per_step": round(ms_median, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{ {'c1': 0, 'c2': 1, 'c3': 2, 'c5': 4}[a.config]}] ({a.config}): " if preset else "custom: ")
                                   + f"{P} synthetic Gaussians (SH deg 3), 1 camera {W}x{H} per GPU, feature_count={fc}, fwd+bwd at the op boundary"
                                   + ("" if world == 1 else (f", cameras on the 8-position ring of SURVEY.md 8(d) (rank r, view v: position r + v x {world}: the views differ in work), " if head == "ring" else
                                                             f", equal-work cameras (the single-GPU camera moved 0.4 degrees per view around the cloud centre; rank r, view v: camera r + v x {world}), ")
                                      + (f"{VPR} views per rank and step whose gradients accumulate, " if VPR > 1 else "one view per rank and step, ")
                                      + f"blocking RCCL sum ({dp_mode}) of the step's gradients (one in-place collective over the gradient arena) at step end")
                                   + (f" [{VPR} views per step, gradients accumulated]" if world == 1 and VPR > 1 else "")
                                   + (f" [ring position {a.ring_position}: NOT the metric's camera]" if world == 1 and a.ring_position else ""),
                       "gaussians": P, "visible": V, "num_rendered": R, "width": W, "height": H, "feature_count": fc,
                       "parallelism": f"view-parallel x{world}", "views_per_rank": VPR},
            "roofline": roof,
            # stage TOTALS per step (a stage with two launches per step -- ranges + quadrant lists -- counts both); each stage is
            # its own event bracket, dispatch latency included: the sum exceeds ms_per_step by a few us per bracket
            "stages_ms": {k: round(v[0] / STAGE_STEPS, 5) for k, v in stages.items() if not k.startswith("unused")},
        }
        if ref_binning is not None:
            out["reference_binning_ms_per_step"] = ref_binning["ms_per_step"]
            out["reference_binning"] = ref_binning
        for key, (msb, nvb) in beside.items():
            out[key + "_ms_per_step"] = round(msb, 4)
            out[key + "_value"] = round(world * nvb * 1e3 / msb, 3)
        if "compute_only" in beside:
            out["scaling_efficiency_vs_compute_only"] = round(beside["compute_only"][0] / ms, 4)  # = value / (what N independent GPUs deliver on this workload)
        out["views_per_step"] = world * VPR
        # the same K steps timed right behind the first W warm-up steps (where rounds 1-3 timed): inside the governor's clock ramp
        out["clock_ramp"] = {"ms_per_step_at_start": round(ms_start, 4), "value_at_start": round(world * VPR * 1e3 / ms_start, 3),
                             "steps_before_headline": "W warm-up + K at-start + reference-binning pass (3 + K) + W warm-up",
                             "note": "sclk reaches 2400 MHz after ~0.5 s of continuous work (tools/clock_trace.py); "
                                     "`value` is timed after that, `value_at_start` before"}
    if world == 1 and rank == 0:
        torch.cuda.empty_cache()
        if not a.no_caller_levels:
            out.update(caller_levels(P, W, H, a.seed, dev))
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, W, H, fc, a.seed)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
