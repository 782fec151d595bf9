"""One training iteration of the reference's geometry stage without the multi-view term (train.py:94-130, 223-227,
258-259) at the bench workload: render(sobel_normal=True) -> clamp -> (1-l)L1 + l(1-SSIM) + plane + depth-normal ->
backward -> densification statistics -> Adam step.  `--reference-formulation` swaps every fused piece for what the
reference itself runs on top of the drop-in rasterizer: PyTorch pre/post-processing in render(), the conv2d `ssim`,
torch.optim.Adam (foreach)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import gs2m_synth as S
import gs2m_optim
from gs2m_scene import GaussianParams, PipelineParams, Camera
from gs2m_losses import l1_loss, plane_loss, depth_normal_loss
import gs2m_losses
from gaussian_renderer import render
from fused_ssim import dssim_loss, fused_ssim

REF = "--reference-formulation" in sys.argv
TORCH_TAIL = REF or "--torch-loss-tail" in sys.argv  # the losses / statistics around the fused pieces as PyTorch expressions
MAT = "--material" in sys.argv  # the material stage (train.py:132-196): deferred PBR shading under a learnable 512^2 environment light
MV = "--multi-view" in sys.argv   # + multi_view_loss against a second, nearby camera (fused path only)
P, W, H = (int(os.environ.get(k, d)) for k, d in (("GS2M_TSB_P", 1_000_000), ("GS2M_TSB_W", 1920), ("GS2M_TSB_H", 1080)))  # small sizes: the host side alone
dev = "cuda"
cam0 = S.make_camera(W, H)
g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam0, seed=0).items()}
albedo = torch.rand(P, 3, device=dev) * 0.8 + 0.1
rough = torch.rand(P, 1, device=dev) * 0.8 + 0.1
metal = torch.rand(P, 1, device=dev) * 0.8 + 0.1
pc = GaussianParams.from_activated(g["means3D"], g["shs"], g["scales"], g["rotations"], g["opacities"].clamp(0.01, 0.99), albedo, rough, metal)
params = [torch.nn.Parameter(t) for t in pc.parameters()]
(pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity, pc._albedo, pc._roughness, pc._metallic) = params
names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic")
lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3, albedo=0.05, roughness=0.05, metallic=0.05)
groups = [{"params": [p], "lr": lrs[n], "name": n} for p, n in zip(params, names)]
opt = (torch.optim.Adam if REF else gs2m_optim.Adam)(groups, lr=0.0, eps=1e-15)
cam = Camera(cam0, dev)
pipe = PipelineParams()
pipe.fused_render_ops = not REF
pipe.split_sh = not REF and "--no-split-sh" not in sys.argv
bg = torch.zeros(3, device=dev)
gt = torch.rand(3, H, W, device=dev)
accum, accum_abs, denom = (torch.zeros(P, 1, device=dev) for _ in range(3))
max_radii = torch.zeros(P, device=dev)
if REF:
    def plane_loss(visibility_filter, gaussians):  # utils/loss_utils.py:72-79 as written there (boolean-mask gather)
        if visibility_filter.sum() == 0:
            return 0.0
        return torch.sort(gaussians.get_scaling[visibility_filter], dim=-1)[0][..., 0].mean()
    from test_ssim_gpu import _window
    import torch.nn.functional as F
    win = _window(3, dev)

    def ssim_fn(a, b):
        mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
        s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1.pow(2)
        s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2.pow(2)
        s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1.pow(2) + mu2.pow(2) + 1e-4) * (s1 + s2 + 9e-4))).mean()
else:
    ssim_fn = fused_ssim


if MV:
    import random
    import gs2m_mvs
    cam_b = Camera(S.look_at_camera(W, H, (0.35, -0.1, 0.0), (0.0, 0.0, 6.0)), dev)
    cam_a = Camera(S.look_at_camera(W, H, (0.0, 0.0, 0.0), (0.0, 0.0, 6.0)), dev)
    cam = cam_a
    mvp = gs2m_mvs.MultiViewParams()
    mvp.multi_view_max_dist = 8.0
    mvs = gs2m_mvs.MultiViewScene([cam_a, cam_b], [gt, torch.rand(3, H, W, device=dev)], pc, mvp)
    rng = random.Random(0)


if MAT:
    import torch.nn.functional as F
    from pbr import CubemapLight, get_brdf_lut, pbr_render
    from gs2m_losses import tv_loss

    class Lighting:
        cubemap = CubemapLight(base_res=512, device=dev)
        brdf_lut = get_brdf_lut().to(dev)
    light_opt = gs2m_optim.Adam([{"name": "cubemap", "params": list(Lighting.cubemap.parameters()), "lr": 0.05}], lr=0.05)
    rays = F.normalize(cam.get_rays().view(-1, 3), p=2, dim=-1)
    TORCH_TV = "--torch-tv" in sys.argv


def material_step():
    """gs2m_train's material-stage iteration (train.py:132-196 without the multi-view roughness term)."""
    global max_radii
    out = render(cam, pc, pipe, bg, geometry_stage=True, material_stage=True, sobel_normal=True)
    vis, radii = out["visibility_filter"], out["radii"]
    loss = gs2m_losses.fused_plane_loss(vis, pc, weight=0.01)
    pkg = pbr_render(Lighting, cam, rays, out, metallic=False)
    if TORCH_TV:
        loss = loss + 0.015 * depth_normal_loss(out["normal_map"], out["sobel_map"], weights=DNW)
        pbr = torch.where(out["normal_mask"], pkg["render_rgb"].permute(2, 0, 1).clamp(0, 1), bg[:, None, None])
        Lpbr = 0.8 * l1_loss(pbr, gt) + 0.2 * (1.0 - fused_ssim(pbr.unsqueeze(0), gt.unsqueeze(0)))
    else:
        pbr, Limg, _ = gs2m_losses.geometry_image_loss(pkg["render_rgb"], gt, out["normal_map"], out["sobel_map"], edge=DNE, w_l1=0.8, w_dn=0.015,
                                                       mask=out["normal_mask"], background=bg)
        Lpbr = Limg + dssim_loss(pbr.unsqueeze(0), gt.unsqueeze(0), 0.2)
    wn = (0.5 * torch.tanh(8.0 * ((1.0 - out["roughness_map"]).detach() - 0.5)) + 0.5).clamp(0, 1)
    if TORCH_TV:
        Lsm = 0.002 * tv_loss(gt, out["roughness_map"], norm1=False) + 0.01 * tv_loss(gt, out["albedo_map"]) + 0.01 * tv_loss(gt, out["normal_map"], weight_map=wn)
    else:
        F_ = gs2m_losses.fused_tv_loss
        Lsm = F_(gt, out["roughness_map"], norm1=False, weight=0.002) + F_(gt, out["albedo_map"], weight=0.01) + F_(gt, out["normal_map"], weight_map=wn, weight=0.01)
    loss = loss + Lpbr + Lsm
    loss.backward()
    with torch.no_grad():
        gs2m_losses.densification_stats(out["viewspace_points"].grad, vis, accum, accum_abs, denom, out["observe"], radii, max_radii)
        opt.step()
        opt.zero_grad(set_to_none=True)
        light_opt.step()
        light_opt.zero_grad(set_to_none=True)
        Lighting.cubemap.clamp_(min=0.0)


def step():
    global max_radii
    if MAT:
        return material_step()
    out = render(cam, pc, pipe, bg, geometry_stage=MV or "--geometry" in sys.argv, material_stage=False, sobel_normal=True)
    image, vis, radii = out["render"], out["visibility_filter"], out["radii"]
    if TORCH_TAIL:
        rgb = image.clamp(0, 1)
        Lssim = 1.0 - ssim_fn(rgb.unsqueeze(0), gt.unsqueeze(0))
        loss = 0.8 * l1_loss(rgb, gt) + 0.2 * Lssim + 0.01 * plane_loss(vis, pc)
        loss = loss + 0.015 * depth_normal_loss(out["normal_map"], out["sobel_map"], gt)
    else:  # the loss tail as fused kernels (csrc/loss_ops.hip)
        rgb, Limg, _ = gs2m_losses.geometry_image_loss(image, gt, out["normal_map"], out["sobel_map"], edge=gs2m_losses.edge_gradient(gt), w_l1=0.8, w_dn=0.015)
        loss = Limg + dssim_loss(rgb.unsqueeze(0), gt.unsqueeze(0), 0.2) + gs2m_losses.fused_plane_loss(vis, pc, weight=0.01)
    if MV:
        loss = loss + gs2m_mvs.multi_view_loss(mvs, cam, mvp, out, pipe, bg, False, render, rng=rng)
    loss.backward()
    with torch.no_grad():  # train.py:223-227, GM:569-573
        if REF:
            mask = (out["observe"] > 0) & vis
            max_radii = torch.where(mask, torch.max(max_radii, radii), max_radii)
            vg = out["viewspace_points"].grad
            accum[vis] += torch.norm(vg[vis, :2], dim=-1, keepdim=True)
            accum_abs[vis] += torch.norm(vg[vis, 2:], dim=-1, keepdim=True)
            denom[vis] += 1
        elif not TORCH_TAIL:
            gs2m_losses.densification_stats(out["viewspace_points"].grad, vis, accum, accum_abs, denom, out["observe"], radii, max_radii)
        else:  # gs2m_model.GaussianModel's masked forms
            mask = (out["observe"] > 0) & vis
            max_radii = torch.where(mask, torch.max(max_radii, radii), max_radii)
            vg = out["viewspace_points"].grad
            f = vis[:, None]
            accum.add_(torch.where(f, torch.norm(vg[:, :2], dim=-1, keepdim=True), 0.0))
            accum_abs.add_(torch.where(f, torch.norm(vg[:, 2:], dim=-1, keepdim=True), 0.0))
            denom.add_(f)
        opt.step()
        opt.zero_grad(set_to_none=True)


if MAT:
    DNW, DNE = gs2m_losses.edge_weights(gt), gs2m_losses.edge_gradient(gt)
for _ in range(5):
    step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print("%s iteration (%s): %.3f ms" % ("material-stage" if MAT else "training", "reference formulation on the drop-in rasterizer" if REF else "fused", (time.perf_counter() - t0) / n * 1e3))
if "--host-profile" in sys.argv:  # where the Python side of an iteration goes (run with a small scene: GS2M_TSB_P=2000 GS2M_TSB_W=64 GS2M_TSB_H=64)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
if "--profile-ops" in sys.argv:  # which framework ops (with shapes) the non-HIP time belongs to
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
    for e in sorted(rows, key=lambda e: -e.self_device_time_total)[:60]:
        print("%-28s n %3d  self device %8.1f us/step  %s" % (e.key, e.count // 3, e.self_device_time_total / 3, str(e.input_shapes)[:110]))
