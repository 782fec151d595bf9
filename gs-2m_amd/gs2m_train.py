"""The reference's training iteration (train.py:76-262) on MI355X: the RGB stage, the geometry stage (with the multi-view
geometric + photometric consistency term when `lambda_multi_view` > 0) and the material stage (with roughness_loss when
`lambda_rough` > 0): render -> clamp ->
(1 - l) L1 + l (1 - SSIM) + plane loss (+ depth-normal loss from `geometry_from_iter`; from `material_from_iter` the
RGB term is replaced by the deferred PBR shading's L1 / D-SSIM plus the edge-aware smoothness terms, and the environment
light gets its own Adam) -> backward -> densification statistics -> densify / prune / opacity reset on the reference's
schedule -> fused Adam step(s).  Every device-side piece is this repository's: the rasterizer, the fused render()
pre/post-processing, fused_ssim, the PBR stage (texture lookups, cubemap prefilters), gs2m_optim.Adam, distCUDA2.

There is no dataset on the GPU box, so the scene is synthetic (SURVEY.md 8(d) C4 says to substitute and say so):
`synthetic_scene()` renders ground-truth views of a known surface-aligned Gaussian object with this rasterizer and
initialises the model from a noisy subsample of its points, the way COLMAP points initialise the reference.

    python gs-2m_amd/gs2m_train.py --iterations 1000 --width 960 --height 540
"""
import argparse
import math
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import torch

import gs2m_synth as S
from fused_ssim import dssim_loss, fused_ssim
from gaussian_renderer import render
from gs2m_losses import (depth_normal_loss, edge_gradient, edge_weights, fused_plane_loss, fused_tv_loss, geometry_image_loss, l1_loss,
                         plane_loss, tv_loss)
from gs2m_model import GaussianModel, OptimizationParams
from gs2m_scene import Camera, GaussianParams, PipelineParams, inverse_sigmoid


def psnr(a, b):
    return (-10.0 * torch.log10(((a - b) ** 2).mean().clamp_min(1e-12))).item()


def synthetic_scene(n_true=60_000, n_views=12, W=640, H=360, init_frac=0.15, seed=0, device="cuda", detail=0.0, splat=0.05,
                    init_noise=0.02, grain=0.0, fx_scale=1.1):
    """-> (cameras, gt_images, init_points, init_colors, cameras_extent)."""
    sc = S.make_surface_scene(n_true, seed=seed, detail=detail, splat=splat, grain=grain)
    t = {k: v.to(device) for k, v in sc.items()}
    truth = GaussianParams(t["points"], t["shs"][:, :1].contiguous(), t["shs"][:, 1:].contiguous(), torch.log(t["scales"]),
                           t["rotations"], inverse_sigmoid(t["opacities"]), *(inverse_sigmoid(torch.full((n_true, c), 0.5, device=device)) for c in (3, 1, 1)))
    cams = [Camera(c, device) for c in S.orbit_cameras(n_views, W, H, radius=6.0, centre=(0.0, -0.8, 6.0), fx=fx_scale * W)]
    pipe, bg = PipelineParams(), torch.zeros(3, device=device)
    with torch.no_grad():
        gts = [render(c, truth, pipe, bg)["render"].clamp(0, 1) for c in cams]
    g = torch.Generator().manual_seed(seed + 1)
    pick = torch.randperm(n_true, generator=g)[: max(1, int(init_frac * n_true))]
    pts = sc["points"][pick] + init_noise * torch.randn(len(pick), 3, generator=g)
    cols = (sc["colors"][pick] + 0.05 * torch.randn(len(pick), 3, generator=g)).clamp(0, 1)
    centers = torch.stack([c.camera_center for c in cams])
    extent = 1.1 * (centers - centers.mean(0)).norm(dim=1).max().item()  # scene/dataset_readers.py getNerfppNorm
    return cams, gts, pts.numpy(), cols.numpy(), extent


class _Lighting:
    """What pbr_render needs of the reference's Scene (scene/__init__.py:44-46, 144-148)."""

    def __init__(self, base_res, lr, device):
        from pbr import CubemapLight, get_brdf_lut
        import gs2m_optim
        self.cubemap = CubemapLight(base_res=base_res, device=device)
        self.brdf_lut = get_brdf_lut().to(device)
        self.light_optimizer = gs2m_optim.Adam([{"name": "cubemap", "params": list(self.cubemap.parameters()), "lr": lr}], lr=lr)


def multi_view_observe_trim(gaussians, cams, pipe, bg, observe_threshold=2):
    """train.py:229-240: prune every Gaussian that fewer than `observe_threshold` training views actually observe (the
    rasterizer's `observe` output counts the pixels a Gaussian contributed to).  -> number of pruned points."""
    with torch.no_grad():
        count = torch.zeros_like(gaussians.get_opacity)
        for view in cams:
            count[render(view, gaussians, pipe, bg, sobel_normal=False)["observe"] > 0] += 1
        prune = (count < observe_threshold).squeeze()
        n = int(prune.sum())
        if n > 0:
            gaussians.prune_points(prune)
    return n


def _view_densification_stats(out, vis, radii):
    """One view's contribution to the densification statistics (train.py:223-227, GM:569-573): the screen-space gradient
    norms where the view sees the Gaussian, the visibility count, and the radius where it both sees and observes it."""
    g = out["viewspace_points"].grad
    f = vis[:, None]
    packed = torch.cat([torch.where(f, torch.norm(g[:, :2], dim=-1, keepdim=True), 0.0),
                        torch.where(f, torch.norm(g[:, 2:], dim=-1, keepdim=True), 0.0), f.to(g.dtype)], dim=1).contiguous()
    mr = torch.where((out["observe"] > 0) & vis, radii, torch.zeros_like(radii)).to(torch.float32)
    return packed, mr


def _reduced_densification_stats(reducer, out, vis, radii, local=None):
    """What a single process that rendered every rank's view(s) would have added to the densification statistics this
    iteration: sums of the per-view screen-space gradient norms over the views that see the Gaussian, the number of such
    views, and the largest radius among the views that both see and observe it.  `local`: the rank's own views already added
    up (views_per_rank > 1); `reducer` None: no other ranks."""
    packed, mr = local if local is not None else _view_densification_stats(out, vis, radii)
    if reducer is not None:
        import torch.distributed as dist
        h1 = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=reducer.group, async_op=True)
        h2 = dist.all_reduce(mr, op=dist.ReduceOp.MAX, group=reducer.group, async_op=True)
        h1.wait(); h2.wait()
    return packed[:, 0:1], packed[:, 1:2], packed[:, 2:3], mr


def export_colmap_dataset(folder, scene):
    """Write a scene (cameras, gt_images, points, colors, extent) as a COLMAP-format dataset the reference's loader
    understands (scene/dataset_readers.py:141-197): `sparse/0/{cameras,images,points3D}.bin` and `images/*.png`."""
    import numpy as np
    from PIL import Image as PILImage
    import gs2m_colmap as C
    cams, gts, pts, cols, _ = scene
    os.makedirs(os.path.join(folder, "images"), exist_ok=True)
    cameras, images = [], []
    for k, (cam, gt) in enumerate(zip(cams, gts)):
        W2C = cam.world_view_transform.transpose(0, 1).double().cpu().numpy()   # rows of the 4x4 world-to-camera matrix
        cameras.append(C.Camera(k + 1, "PINHOLE", cam.image_width, cam.image_height,
                                np.array([cam.Fx, cam.Fy, 0.5 * cam.image_width, 0.5 * cam.image_height], dtype=np.float64)))
        name = f"view_{k:03d}.png"
        images.append(C.Image(k + 1, C.rotmat2qvec(W2C[:3, :3]), W2C[:3, 3], k + 1, name, None, None))
        arr = (gt.clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
        PILImage.fromarray(arr).save(os.path.join(folder, "images", name))
    C.write_model(os.path.join(folder, "sparse", "0"), cameras, images, np.asarray(pts, dtype=np.float64),
                  (np.asarray(cols) * 255.0 + 0.5).astype(np.uint8))


def load_colmap_dataset(folder, images="images", device="cuda", resolution=1):
    """COLMAP-format dataset -> the `scene` tuple `train()` takes: cameras with the reference's matrices
    (scene/cameras.py:58-67), ground-truth images, the sparse points and colours, and the scene radius
    (readColmapSceneInfo, scene/dataset_readers.py:141-197; images sorted by name, PINHOLE / SIMPLE_PINHOLE only)."""
    import numpy as np
    from PIL import Image as PILImage
    import gs2m_colmap as C
    sparse = os.path.join(folder, "sparse", "0")
    intr = C.read_intrinsics_binary(os.path.join(sparse, "cameras.bin"))
    infos = sorted(C.colmap_cameras(C.read_extrinsics_binary(os.path.join(sparse, "images.bin")), intr), key=lambda c: c.image_name)
    cams, gts = [], []
    for info in infos:
        img = PILImage.open(os.path.join(folder, images, info.image_name)).convert("RGB")
        assert img.size == (info.width, info.height), "image size differs from the COLMAP camera"
        w, h, fx, fy = info.width, info.height, info.Fx, info.Fy
        if resolution != 1:  # `-r 2`: images and intrinsics scaled down together (utils/camera_utils.py loadCam)
            w, h = round(info.width / resolution), round(info.height / resolution)
            img = img.resize((w, h), PILImage.LANCZOS)
            fx, fy = fx * w / info.width, fy * h / info.height
        gts.append(torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous().to(device))
        cams.append(Camera(S.make_camera(w, h, fx=fx, fy=fy, R=info.R, T=info.T), device))
    xyz, rgb, _ = C.read_points3D_binary(os.path.join(sparse, "points3D.bin"))
    return cams, gts, xyz.astype(np.float32), (rgb / 255.0).astype(np.float32), C.nerf_normalization(infos)["radius"]


# ---- BASELINE.json configs[3] ("C4"): DTU scan24 through the full train.py loop --------------------------------------
# scripts/run_dtu.py:21-22 runs `train.py -s dtu/scan24 -r 2 --lambda_depth_normal 0.015`: 49 images of 1554 x 1162 read at
# half resolution (777 x 581), ~30 k COLMAP points densified past 300 k, geometry stage with the multi-view term.  The
# dataset is not on the GPU box (no network), so the scene is a SUBSTITUTE, and every line that reports it says so: a
# synthetic object of `n_true` surface-aligned Gaussians with a fine colour texture, rendered by this rasterizer from 49
# orbit cameras at 1554 x 1162, WRITTEN as a COLMAP-format dataset (sparse/0/*.bin + images/*.png) and read back through
# the COLMAP loader at `-r 2` exactly as the reference reads a scan.
C4 = dict(n_views=49, full_W=1554, full_H=1162, resolution=2, n_true=1_500_000, init_points=30_000, detail=60.0, splat=0.008, grain=0.4,
          fx_scale=3.0, lambda_depth_normal=0.015)


def c4_scene(folder, n_true=None, init_points=None, seed=0, device="cuda"):
    """-> the `scene` tuple of `train()` for the C4 substitute, through a COLMAP-format dataset written to `folder`."""
    n_true = n_true or C4["n_true"]
    init_points = init_points or C4["init_points"]
    full = synthetic_scene(n_true, C4["n_views"], C4["full_W"], C4["full_H"], init_frac=init_points / n_true, seed=seed, device=device,
                           detail=C4["detail"], splat=C4["splat"], init_noise=0.01, grain=C4["grain"], fx_scale=C4["fx_scale"])
    export_colmap_dataset(folder, full)
    del full
    if torch.device(device).type == "cuda":
        torch.cuda.empty_cache()
    return load_colmap_dataset(folder, resolution=C4["resolution"], device=device)


def c4_options(iterations):
    """The reference's schedule (arguments/__init__.py:81-134: densify 500-15000 every 100, opacity reset every 3000, geometry
    stage from 5000 of 30000 iterations) compressed onto `iterations` with the same proportions, `--lambda_depth_normal 0.015`
    as scripts/run_dtu.py:21 passes it.  -> (OptimizationParams, geometry_from_iter, MultiViewParams)"""
    import gs2m_mvs
    opt = OptimizationParams()
    k = iterations / 30_000.0
    opt.lambda_depth_normal = C4["lambda_depth_normal"]
    opt.densify_from_iter = max(100, int(round(500 * k)))
    opt.densify_until_iter = int(round(15_000 * k))
    opt.densification_interval = max(25, int(round(100 * k)))
    opt.opacity_reset_interval = max(200, int(round(3000 * k)))
    opt.position_lr_max_steps = iterations
    mv = gs2m_mvs.MultiViewParams()
    # 49 orbit cameras 6 units from the object: neighbours 7.3 degrees / 0.77 units apart (DTU's cameras sit ~0.1-0.2 of
    # the scene radius apart: the reference's 1.5 / 30 degrees default bounds already admit them)
    return opt, int(round(5000 * k)), mv


def c4_run(folder, iterations=5000, seed=0, callback=None, log=None, multi_view=True, n_true=None, init_points=None, scene=None,
           schedule_iterations=None):
    """One C4 training run on the substitute scene: RGB stage, then the geometry stage with the depth-normal term at 0.015 and
    the multi-view consistency term, densify / prune / opacity reset / observe trim on the reference's (compressed) schedule.
    `schedule_iterations`: the run length the schedule is laid out for when only its first `iterations` are run (a test that
    repeats the beginning of a longer run).  -> (model, stats); stats also carries `points_max` and the scene's description."""
    import random
    scene = scene or c4_scene(folder, n_true=n_true, init_points=init_points, seed=seed)
    opt, geometry_from, mv = c4_options(schedule_iterations or iterations)
    random.seed(seed)  # the multi-view term picks its neighbour view with `random` (utils/loss_utils.py multi_view_loss)
    peak = [0]

    def cb(it, g, cams, gts):
        peak[0] = max(peak[0], g.get_xyz.shape[0])
        if callback is not None:
            callback(it, g, cams, gts)

    model, st = train(iterations=iterations, geometry_from_iter=geometry_from, opt=opt, scene=scene, seed=seed, log=log,
                      lambda_multi_view=OptimizationParams.lambda_multi_view if multi_view else 0.0, mv_opt=mv,
                      trim_interval=max(250, (schedule_iterations or iterations) // 4), callback=cb)
    st["points_max"] = peak[0]
    st["workload"] = (f"C4 SUBSTITUTE for DTU scan24 (dataset absent): synthetic COLMAP-format scene, {len(scene[0])} views "
                      f"{scene[0][0].image_width}x{scene[0][0].image_height} (written at {C4['full_W']}x{C4['full_H']}, read at -r {C4['resolution']}), "
                      f"{st['points_start']} initial points, geometry stage from iteration {geometry_from} with lambda_depth_normal "
                      f"{C4['lambda_depth_normal']}" + (" and the multi-view term" if multi_view else "") + f", {iterations} iterations")
    return model, st


def load_blender_dataset(folder, transforms="transforms_train.json", extension=".png", white_background=False, n_points=100_000, seed=0,
                         device="cuda"):
    """NeRF-synthetic ("Blender") dataset -> the `scene` tuple `train()` takes (readCamerasFromTransforms / readNerfSyntheticInfo,
    scene/dataset_readers.py:199-274): `camera_angle_x` + per-frame camera-to-world matrices in OpenGL axes (y up, z back), flipped
    to COLMAP axes, inverted, rotation stored transposed; RGBA images composited on the background; no sparse points, so a random
    cloud in [-1.3, 1.3]^3 with near-black colours as there."""
    import json
    import numpy as np
    from PIL import Image as PILImage
    import gs2m_colmap as C
    with open(os.path.join(folder, transforms)) as f:
        meta = json.load(f)
    cams, gts, infos = [], [], []
    for idx, frame in enumerate(meta["frames"]):
        c2w = np.array(frame["transform_matrix"], dtype=np.float64)
        c2w[:3, 1:3] *= -1
        w2c = np.linalg.inv(c2w)
        R, T = np.transpose(w2c[:3, :3]), w2c[:3, 3]
        img = PILImage.open(os.path.join(folder, frame["file_path"] + extension))
        rgba = np.asarray(img.convert("RGBA"), dtype=np.float32) / 255.0
        bgc = 1.0 if white_background else 0.0
        rgb = rgba[..., :3] * rgba[..., 3:4] + bgc * (1.0 - rgba[..., 3:4])
        W, H = img.size
        focal = W / (2.0 * math.tan(meta["camera_angle_x"] / 2.0))            # fov2focal
        gts.append(torch.from_numpy(rgb).permute(2, 0, 1).contiguous().to(device))
        cams.append(Camera(S.make_camera(W, H, fx=focal, fy=focal, R=R, T=T), device))
        infos.append(C.CameraInfo(idx, R, T, focal, focal, os.path.basename(frame["file_path"]) + extension, W, H))
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n_points, 3)) * 2.6 - 1.3).astype(np.float32)
    cols = (rng.random((n_points, 3)) / 255.0 * 0.28209479177387814 + 0.5).astype(np.float32)   # SH2RGB of near-zero coefficients
    return cams, gts, xyz, cols, C.nerf_normalization(infos)["radius"]


def train(iterations=1000, W=640, H=360, n_views=12, n_true=60_000, seed=0, geometry_from_iter=None, opt=None, log=None,
          device="cuda", scene=None, material_from_iter=None, light_res=128, lambda_smooth=0.0, lambda_normal=0.1,
          lambda_multi_view=0.0, mv_opt=None, lambda_rough=0.0, trim_interval=1000, alpha_masks=None,
          white_background=False, dp=False, dp_mode="auto", ssim_fn=None, optimizer_cls=None, pipe=None, views_per_rank=1,
          callback=None):
    """`dp=True`: view-parallel data parallelism over the initialised torch.distributed group: every
    rank holds the full model, an iteration renders `world_size` different views (rank r takes the r-th of the next
    `world_size` entries of the shared random view order), the parameter gradients are SUMMED over the ranks with one
    blocking collective before the optimizer step, and the densification side channels are reduced with the
    single-process semantics (per-view gradient norms and visibility counts summed, radii max-reduced, train.py:223-227,
    scene/gaussian_model.py:569-573).  Every rank then takes the same densify / prune / reset decisions -- the
    generator behind densify_and_split's torch.normal is seeded identically and consumed identically -- so the
    replicas stay bit-identical (tests/test_dp.py).  Schedules (densification, resets, stages) count iterations, i.e.
    one iteration consumes `world_size` x `views_per_rank` views.
    `views_per_rank` > 1 (with or without dp): gradient ACCUMULATION over that many views per rank and iteration -- each
    backward adds to the gradients, the densification statistics of the rank's views are added up locally, one reduction
    follows the last view and the optimizer steps once: the sums are exact (no gradient is applied a step late) and the
    collective is paid once per `views_per_rank` views, which is what lets the view-parallel form scale past the point
    where one view's gradients cost as much wire time as the view costs compute.
    `dp_mode`: "allreduce", "rs_ag" (reduce-scatter + all-gather: every link of the xGMI mesh busy) or "auto" = rs_ag from 4
    ranks on.
    `callback(it, gaussians, cams, gts)`: called under no_grad at the end of every iteration (after the optimizer step): a
    test's window on the run (mid-run parity spot checks, snapshots for determinism checks)."""
    opt = opt or OptimizationParams()
    ssim = ssim_fn or fused_ssim
    rank, world, reducer = 0, 1, None
    if dp:
        import torch.distributed as dist
        from gs2m_dp import GradReducer
        assert dist.is_initialized(), "train(dp=True) needs an initialised torch.distributed process group"
        rank, world = dist.get_rank(), dist.get_world_size()
        reducer = GradReducer(mode=("rs_ag" if world >= 4 else "allreduce") if dp_mode == "auto" else dp_mode)
        import gs2m_arena
        gs2m_arena.set_keep(views_per_rank + 1)  # the first view's arena (the accumulated gradients) stays registered until the reduction
    geometry_from_iter = iterations // 2 if geometry_from_iter is None else geometry_from_iter
    material_from_iter = iterations + 1 if material_from_iter is None else material_from_iter
    cams, gts, pts, cols, extent = scene or synthetic_scene(n_true, n_views, W, H, seed=seed, device=device)
    torch.manual_seed(seed)
    gaussians = GaussianModel(3, device)
    gaussians.create_from_pcd(pts, cols, extent)
    gaussians.training_setup(opt, **({"optimizer_cls": optimizer_cls} if optimizer_cls else {}))
    pipe, bg = pipe or PipelineParams(), torch.zeros(3, device=device)

    def evaluate():
        with torch.no_grad():
            return sum(psnr(render(c, gaussians, pipe, bg)["render"].clamp(0, 1), gt) for c, gt in zip(cams, gts)) / len(cams)

    mv_scene = None
    if lambda_multi_view > 0 or lambda_rough > 0:  # train.py:121-130, 195: the multi-view terms
        import gs2m_mvs
        mv_opt = mv_opt or gs2m_mvs.MultiViewParams()
        mv_scene = gs2m_mvs.MultiViewScene(cams, gts, gaussians, mv_opt)
    lighting, rays, dn_weights, dn_edges = None, {}, {}, {}
    if material_from_iter < iterations:
        from pbr import pbr_render
        import torch.nn.functional as F
        lighting = _Lighting(light_res, opt.opacity_lr, device)
    stats = dict(psnr_start=evaluate(), points_start=gaussians.get_xyz.shape[0], pbr_loss=[])
    order = torch.Generator().manual_seed(seed)
    stack = []
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    # Python's cyclic collector walks every tracked object of the process on a full collection (36 ms with the ~170 k objects torch's
    # import leaves; tools/stall_probe.py), at moments of its own choosing: the set-up state goes to the permanent generation, so the
    # collections inside the loop only look at what the loop creates.
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    for it in range(1, iterations + 1):
        gaussians.update_learning_rate(it)
        if it % 1000 == 0:
            gaussians.oneupSHdegree()
        if not stack:
            stack = torch.randperm(len(cams), generator=order).tolist()
        # the views of this iteration: `views_per_rank` per rank (1: the reference's loop, train.py:86-97).  With more than one,
        # every view's backward ADDS to the parameters' gradients (autograd's accumulation), the per-view densification
        # statistics are added up locally, and ONE reduction per iteration follows the last view: exact gradient accumulation,
        # no stale gradients -- the collective's cost is paid once per `views_per_rank` views.
        need = world * views_per_rank
        while len(stack) < need:
            stack = torch.randperm(len(cams), generator=order).tolist() + stack
        mine = [stack.pop() for _ in range(need)]
        my_views = mine[rank * views_per_rank:(rank + 1) * views_per_rank]  # rank r: a contiguous block of the shared order
        geometry_stage = it > geometry_from_iter
        material_stage = it > material_from_iter

        def view_step(k):
            """forward, losses and backward of view k (train.py:98-217); gradients accumulate in the parameters' .grad"""
            cam, gt = cams[k], gts[k]
            out = render(cam, gaussians, pipe, bg, geometry_stage, material_stage, sobel_normal=geometry_stage)
            vis, radii = out["visibility_filter"], out["radii"]
            # the loss tail as fused kernels (gs2m_losses, csrc/loss_ops.hip) on the GPU; the PyTorch expressions otherwise
            fused_tail = out["render"].is_cuda and getattr(pipe, "fused_loss_tail", True)
            fused_image = fused_tail and not material_stage
            rgb = None if fused_image else out["render"].clamp(0, 1)
            loss = fused_plane_loss(vis, gaussians, weight=opt.lambda_plane) if fused_tail else opt.lambda_plane * plane_loss(vis, gaussians)
            if alpha_masks is not None:  # train.py:108-109: opacity against the foreground mask (white-background / masked datasets)
                loss = loss + opt.lambda_alpha * torch.nn.functional.binary_cross_entropy(out["alpha_map"].clamp(0.0, 1.0), alpha_masks[k])
            if fused_image:  # train.py:101-120 in one pass: clamp, L1 and the edge-weighted depth-normal term
                if geometry_stage and k not in dn_edges:  # a function of the ground-truth image only (the reference recomputes it every iteration)
                    dn_edges[k] = edge_gradient(gt)
                rgb, Limg, _ = geometry_image_loss(out["render"], gt, out["normal_map"] if geometry_stage else None,
                                                   out["sobel_map"] if geometry_stage else None, edge=dn_edges[k] if geometry_stage else None,
                                                   w_l1=1.0 - opt.lambda_ssim, w_dn=opt.lambda_depth_normal if geometry_stage else 0.0)
                if ssim_fn is None:  # lambda (1 - ssim) as one node
                    loss = loss + Limg + dssim_loss(rgb.unsqueeze(0), gt.unsqueeze(0), opt.lambda_ssim)
                else:
                    loss = loss + Limg + opt.lambda_ssim * (1.0 - ssim(rgb.unsqueeze(0), gt.unsqueeze(0)))
            elif not material_stage:  # train.py:101-115
                Lssim = 1.0 - ssim(rgb.unsqueeze(0), gt.unsqueeze(0))
                loss = loss + (1.0 - opt.lambda_ssim) * l1_loss(rgb, gt) + opt.lambda_ssim * Lssim
            if geometry_stage:
                if fused_tail and material_stage and ssim_fn is None:
                    pass  # rides along with the shaded image's L1 below (one pass over the frame)
                elif not fused_image:
                    if k not in dn_weights:
                        dn_weights[k] = edge_weights(gt)
                    loss = loss + opt.lambda_depth_normal * depth_normal_loss(out["normal_map"], out["sobel_map"], weights=dn_weights[k])
                if mv_scene is not None and lambda_multi_view > 0:
                    Lmv = gs2m_mvs.multi_view_loss(mv_scene, cam, mv_opt, out, pipe, bg, material_stage, render,
                                                   fused=os.environ.get("GS2M_MV_OP_BY_OP") is None)  # debugging aid: the op-by-op formulation
                    loss = loss + lambda_multi_view * Lmv
                    # (kept on the device: a float() here would make the host wait for the whole forward before it queues the backward)
                    stats.setdefault("mv_loss", []).append(Lmv.detach() if torch.is_tensor(Lmv) else float(Lmv))
            if material_stage:  # train.py:132-196
                if k not in rays:
                    rays[k] = F.normalize(cam.get_rays().view(-1, 3), p=2, dim=-1)
                pkg = pbr_render(lighting, cam, rays[k], out, metallic=False)
                if fused_tail and ssim_fn is None:
                    # where(normal_mask, clamp(render_rgb^T, 0, 1), bg), its L1 to the ground truth and (geometry stage) the
                    # depth-normal term in ONE pass -- the shading's (H,W,3) output goes in as it is -- and D-SSIM as one node
                    if geometry_stage and k not in dn_edges:
                        dn_edges[k] = edge_gradient(gt)
                    pbr, Limg, terms = geometry_image_loss(pkg["render_rgb"], gt, out["normal_map"] if geometry_stage else None,
                                                           out["sobel_map"] if geometry_stage else None, edge=dn_edges[k] if geometry_stage else None,
                                                           w_l1=1.0 - opt.lambda_ssim, w_dn=opt.lambda_depth_normal if geometry_stage else 0.0,
                                                           mask=out["normal_mask"], background=bg)
                    Lds = dssim_loss(pbr.unsqueeze(0), gt.unsqueeze(0), opt.lambda_ssim)
                    Lpbr = Limg + Lds                                               # in the graph (carries the depth-normal term too)
                    Lpbr_log = (1.0 - opt.lambda_ssim) * terms[0] + Lds.detach()    # the reference's Lpbr, for the statistics
                else:
                    pbr = torch.where(out["normal_mask"], pkg["render_rgb"].permute(2, 0, 1).clamp(0, 1), bg[:, None, None])
                    Lpbr = (1.0 - opt.lambda_ssim) * l1_loss(pbr, gt) + opt.lambda_ssim * (1.0 - ssim(pbr.unsqueeze(0), gt.unsqueeze(0)))
                    Lpbr_log = Lpbr
                wn = (0.5 * torch.tanh(8.0 * ((1.0 - out["roughness_map"]).detach() - 0.5)) + 0.5).clamp(0, 1)
                if fused_tail:  # the lambdas folded into the nodes
                    Lsm = (fused_tv_loss(gt, out["roughness_map"], norm1=False, weight=lambda_smooth) + fused_tv_loss(gt, out["albedo_map"], weight=0.01)
                           + fused_tv_loss(gt, out["normal_map"], weight_map=wn, weight=lambda_normal))
                else:
                    Lsm = (lambda_smooth * tv_loss(gt, out["roughness_map"], norm1=False) + 0.01 * tv_loss(gt, out["albedo_map"])
                           + lambda_normal * tv_loss(gt, out["normal_map"], weight_map=wn))
                loss = loss + Lpbr + Lsm
                if mv_scene is not None and lambda_rough > 0:  # train.py:194-195
                    loss = loss + lambda_rough * gs2m_mvs.roughness_loss(mv_scene, cam, mv_opt, out, pipe, bg, render)
                stats["pbr_loss"].append(Lpbr_log.detach())  # (on the device until the run ends: no host wait per iteration)
            loss.backward()
            return loss, out, vis, radii, fused_tail

        local = None  # views_per_rank > 1: this rank's densification statistics, added up view by view
        for k in my_views:
            loss, out, vis, radii, fused_tail = view_step(k)
            if views_per_rank > 1 and it <= opt.densify_until_iter:
                with torch.no_grad():
                    pk, mr = _view_densification_stats(out, vis, radii)
                    local = (pk, mr) if local is None else (local[0] + pk, torch.max(local[1], mr))
        with torch.no_grad():
            # ---- train.py:219-254, in the reference's order: densification statistics and densify / prune, the
            # multi-view observe trim, THEN the opacity reduce / reset (a trim that ran after a reset would count
            # `observe` with every opacity at 0.01) ----
            if dp:  # the sum of the ranks' parameter gradients (gs2m_dp): in place where autograd left them in a gradient arena
                # (SH straight from the rasterizer, the six raw-parameter gradients of the fused activation backward), only
                # the active SH bands below the maximal degree (train.py:81-82)
                n_act = (gaussians.active_sh_degree + 1) ** 2
                sh_act = [(gaussians._features_rest, n_act - 1)] if gaussians.active_sh_degree < gaussians.max_sh_degree else None
                reducer.reduce_parameter_grads([g["params"][0] for g in gaussians.optimizer.param_groups], sh_active=sh_act)
                if lighting is not None and material_stage:
                    reducer.reduce_parameter_grads(list(lighting.cubemap.parameters()))
            if it <= opt.densify_until_iter:
                if dp or views_per_rank > 1:
                    gn, ga, cnt, mr = _reduced_densification_stats(reducer if dp else None, out, vis, radii, local)
                    gaussians.max_radii2D = torch.max(gaussians.max_radii2D, mr)
                    gaussians.xyz_gradient_accum += gn
                    gaussians.xyz_gradient_accum_abs += ga
                    gaussians.denom += cnt
                else:
                    gaussians.accumulate_view_stats(out["viewspace_points"], vis, out["observe"], radii, fused=fused_tail)
                if it > opt.densify_from_iter and it % opt.densification_interval == 0:
                    thr = opt.radii2D_threshold if it > opt.opacity_reset_interval else None
                    gaussians.densify_and_prune(opt.densify_grad_threshold, opt.densify_grad_abs_threshold, opt.opacity_prune_threshold, extent, thr)
            if opt.use_multi_view_trim and trim_interval and it % trim_interval == 0 and it < opt.densify_until_iter:
                stats["trimmed"] = stats.get("trimmed", 0) + multi_view_observe_trim(gaussians, cams, pipe, bg)
            if it <= opt.densify_until_iter:
                if opt.use_opacity_reduce and it % opt.opacity_reduce_interval == 0:  # train.py:243-246
                    gaussians.reduce_opacity()
                if it % opt.opacity_reset_interval == 0 or (white_background and it == opt.densify_from_iter):  # train.py:248-250
                    gaussians.reset_opacity()
            if it < iterations:
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
                if material_stage:  # train.py:260-263
                    lighting.light_optimizer.step()
                    lighting.light_optimizer.zero_grad(set_to_none=True)
                    lighting.cubemap.clamp_(min=0.0)
        if callback is not None:
            with torch.no_grad():
                callback(it, gaussians, cams, gts)
        if log and it % log == 0:
            print(f"[{it:6d}] loss {loss.item():.5f}  points {gaussians.get_xyz.shape[0]}", flush=True)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for name in ("mv_loss", "pbr_loss"):  # the per-iteration loss values waited on the device
        if stats.get(name):
            stats[name] = [float(v) for v in stats[name]]
    stats.update(psnr_end=evaluate(), points_end=gaussians.get_xyz.shape[0], seconds=dt, it_per_s=iterations / dt, loss_end=loss.item())
    if lighting is not None:  # the material stage supervises the PBR image, not the SH colours (train.py:111-113): report that one too
        with torch.no_grad():
            tot = 0.0
            for k, (c, gt) in enumerate(zip(cams, gts)):
                out = render(c, gaussians, pipe, bg, True, True)
                if k not in rays:
                    rays[k] = F.normalize(c.get_rays().view(-1, 3), p=2, dim=-1)
                pkg = pbr_render(lighting, c, rays[k], out, metallic=False)
                tot += psnr(torch.where(out["normal_mask"], pkg["render_rgb"].permute(2, 0, 1).clamp(0, 1), bg[:, None, None]), gt)
            stats["psnr_pbr_end"] = tot / len(cams)
    stats["lighting"] = lighting
    return gaussians, stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=1000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--true-gaussians", type=int, default=60_000)
    ap.add_argument("--save-ply", default=None)
    ap.add_argument("--geometry-from", type=int, default=None, help="iteration the geometry stage starts at (default: half way)")
    ap.add_argument("--material-from", type=int, default=None, help="iteration the material stage starts at (default: never)")
    ap.add_argument("--multi-view", action="store_true", help="multi_view_loss in the geometry stage, roughness_loss in the material stage")
    ap.add_argument("--source-path", "-s", default=None, help="COLMAP-format dataset (sparse/0/*.bin + images/); default: synthetic scene")
    ap.add_argument("--resolution", "-r", type=int, default=1, help="down-scale factor for a COLMAP dataset's images")
    ap.add_argument("--blender", action="store_true", help="--source-path is a NeRF-synthetic (transforms_train.json) dataset")
    ap.add_argument("--export-colmap", default=None, help="write the synthetic scene as a COLMAP-format dataset to this folder and exit")
    ap.add_argument("--c4", default=None, metavar="FOLDER", help="BASELINE configs[3] substitute (see C4 above): write the scene to FOLDER, train --iterations on it")
    ap.add_argument("--c4-detail", type=float, default=None, help="experiment: texture frequency of the C4 substitute scene")
    ap.add_argument("--c4-true", type=int, default=None, help="experiment: true Gaussians of the C4 substitute scene")
    ap.add_argument("--c4-splat", type=float, default=None)
    ap.add_argument("--c4-grain", type=float, default=None)
    ap.add_argument("--c4-fx", type=float, default=None)
    a = ap.parse_args()
    if a.c4:
        if a.c4_detail is not None:
            C4["detail"] = a.c4_detail
        if a.c4_true is not None:
            C4["n_true"] = a.c4_true
        if a.c4_splat is not None:
            C4["splat"] = a.c4_splat
        if a.c4_grain is not None:
            C4["grain"] = a.c4_grain
        if a.c4_fx is not None:
            C4["fx_scale"] = a.c4_fx
        model, st = c4_run(a.c4, iterations=a.iterations, log=max(1, a.iterations // 20))
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items() if k not in ("pbr_loss", "lighting", "mv_loss")})
        sys.exit(0)
    if a.export_colmap:
        export_colmap_dataset(a.export_colmap, synthetic_scene(a.true_gaussians, a.views, a.width, a.height))
        sys.exit(0)
    scene = None
    if a.source_path:
        scene = load_blender_dataset(a.source_path) if a.blender else load_colmap_dataset(a.source_path, resolution=a.resolution)
    mv = None
    if a.multi_view:
        import gs2m_mvs
        mv = gs2m_mvs.MultiViewParams()
        if scene is None:  # the synthetic orbit: cameras 6 units from the object, 30 degrees apart
            mv.multi_view_max_dist, mv.multi_view_max_angle, mv.nearby_cam_max_dist = 8.0, 35, 8.0
    model, st = train(a.iterations, a.width, a.height, a.views, a.true_gaussians, log=max(1, a.iterations // 10), scene=scene,
                      geometry_from_iter=a.geometry_from, material_from_iter=a.material_from, mv_opt=mv,
                      lambda_multi_view=OptimizationParams.lambda_multi_view if a.multi_view else 0.0,
                      lambda_rough=OptimizationParams.lambda_rough if a.multi_view else 0.0)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items() if k not in ("pbr_loss", "lighting", "mv_loss")})
    if a.save_ply:
        model.save_ply(a.save_ply)
