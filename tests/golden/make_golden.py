"""Generates the committed golden vectors.  Run in the BUILD container only:

    python tests/golden/make_golden.py

(1) ref_helpers.npz -- outputs of the reference's own pure-Python helpers, imported from
    /root/reference (they cannot travel to the GPU box, the vectors can):
      utils/sh_utils.py:eval_sh (:57-112), utils/graphics_utils.py:getWorld2View2 /
      getProjectionMatrix (:38-71), utils/normal_utils.py:normal_from_depth_image (:65-72).
    These pin the oracle's SH forward, the synthetic-camera matrices and render()'s Sobel path.
(2) raster_small.npz -- inputs and the CPU oracle's outputs/gradients for one small scene; a
    regression anchor for the oracle and a fixed vector for the HIP path.  (The reference has no
    runnable rasterizer here -- CUDA only -- so this file is NOT reference output; it says so.)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def ref_helpers():
    sys.path.insert(0, "/root/reference")
    from utils.sh_utils import eval_sh
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix
    from utils.normal_utils import normal_from_depth_image
    g = torch.Generator().manual_seed(1234)
    out = {}
    # SH: (P, 3, 16) coefficients, unit directions
    sh = torch.randn(257, 3, 16, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(257, 3, generator=g), dim=1)
    out["sh_coeffs"] = sh.numpy()
    out["sh_dirs"] = dirs.numpy()
    for deg in range(4):
        out[f"sh_eval_deg{deg}"] = eval_sh(deg, sh, dirs).numpy()
    # cameras
    Rs, Ts, views, projs = [], [], [], []
    for i in range(6):
        q = torch.nn.functional.normalize(torch.randn(4, generator=g), dim=0).numpy().astype(np.float64)
        r, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        T = torch.randn(3, generator=g).numpy().astype(np.float64) * 3
        fovx, fovy = 0.6 + 0.1 * i, 0.4 + 0.07 * i
        Rs.append(R); Ts.append(T)
        views.append(getWorld2View2(R, T))
        projs.append(getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).numpy())
    out["cam_R"] = np.stack(Rs); out["cam_T"] = np.stack(Ts)
    out["cam_fov"] = np.array([[0.6 + 0.1 * i, 0.4 + 0.07 * i] for i in range(6)])
    out["cam_view"] = np.stack(views); out["cam_proj"] = np.stack(projs)
    # normals from depth
    depth = 2.0 + torch.rand(37, 53, generator=g)
    K = torch.tensor([[60.0, 0, 26.5], [0, 55.0, 18.5], [0, 0, 1]])
    E = torch.tensor(views[0])
    out["nd_depth"] = depth.numpy(); out["nd_K"] = K.numpy(); out["nd_E"] = E.numpy()
    out["nd_world"] = normal_from_depth_image(depth, K, E, view_space=False).numpy()
    out["nd_view"] = normal_from_depth_image(depth, K, E, view_space=True).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_helpers.npz"), **out)
    print("wrote ref_helpers.npz")


def ref_model_helpers():
    """(3) ref_model.npz -- the reference's host-side helpers behind GaussianModel (importable here: pure Python):
    utils/general_utils.py:get_expon_lr_func (:36-66), inverse_sigmoid (:20-21); utils/sh_utils.py:RGB2SH (:114-115)."""
    sys.path.insert(0, "/root/reference")
    from utils.general_utils import get_expon_lr_func, inverse_sigmoid
    from utils.sh_utils import RGB2SH
    out = {}
    steps = np.array([0, 1, 7, 100, 999, 1000, 5000, 15000, 29999, 30000, 40000])
    out["lr_steps"] = steps
    cfgs = [(0.00016 * 6.6, 0.0000016 * 6.6, 0, 0.01, 30000), (1e-2, 1e-4, 1000, 0.01, 20000), (3e-3, 3e-3, 0, 1.0, 100)]
    out["lr_cfgs"] = np.array(cfgs, dtype=np.float64)
    out["lr_values"] = np.array([[get_expon_lr_func(lr_init=a, lr_final=b, lr_delay_steps=int(c), lr_delay_mult=d, max_steps=int(e))(int(s))
                                  for s in steps] for a, b, c, d, e in cfgs], dtype=np.float64)
    g = torch.Generator().manual_seed(99)
    x = torch.rand(64, 3, generator=g) * 0.98 + 0.01
    out["unit_x"] = x.numpy()
    out["inverse_sigmoid"] = inverse_sigmoid(x).numpy()
    out["rgb2sh"] = RGB2SH(x).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_model.npz"), **out)
    print("wrote ref_model.npz")


def ref_defaults():
    """(5) ref_defaults.json -- the reference's hyper-parameter defaults (arguments/__init__.py: OptimizationParams,
    PipelineParams), read from the classes themselves."""
    import json
    from argparse import ArgumentParser
    sys.path.insert(0, "/root/reference")
    import arguments
    p = ArgumentParser()
    out = {"OptimizationParams": {k: v for k, v in vars(arguments.OptimizationParams(p)).items() if not k.startswith("_")},
           "PipelineParams": {k: v for k, v in vars(arguments.PipelineParams(p)).items() if not k.startswith("_")}}
    json.dump(out, open(os.path.join(HERE, "ref_defaults.json"), "w"), indent=1, sort_keys=True)
    print("wrote ref_defaults.json")


def colmap_small():
    """(4) colmap_small/ -- a small synthetic COLMAP binary model (written by gs2m_colmap.write_model: the files are test
    DATA) and colmap_small.npz -- what the REFERENCE's reader (scene/colmap_loader.py:123-240, loaded as a standalone
    module: scene/__init__ needs plyfile) and camera conventions (qvec2rotmat :41-51, getNerfppNorm's arithmetic via
    utils/graphics_utils.getWorld2View2) return for them."""
    import importlib.util
    import gs2m_colmap as C
    rng = np.random.default_rng(7)
    folder = os.path.join(HERE, "colmap_small")
    cams = [C.Camera(1, "PINHOLE", 640, 360, np.array([700.5, 701.25, 320.0, 180.0])),
            C.Camera(2, "SIMPLE_PINHOLE", 800, 600, np.array([910.0, 400.0, 300.0]))]
    images = []
    for k in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if q[0] < 0: q = -q
        m = int(rng.integers(0, 6))
        images.append(C.Image(10 + k, q, rng.normal(size=3) * 3, 1 + k % 2, f"rect_{k:03d}_3_r5000.png",
                              rng.uniform(0, 600, size=(m, 2)), rng.integers(-1, 40, size=m)))
    xyz = rng.normal(size=(23, 3)) * 2
    rgb = rng.integers(0, 256, size=(23, 3))
    err = rng.uniform(0, 2, size=23)
    C.write_model(folder, cams, images, xyz, rgb, err)
    spec = importlib.util.spec_from_file_location("ref_colmap_loader", "/root/reference/scene/colmap_loader.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    sys.path.insert(0, "/root/reference")
    from utils.graphics_utils import getWorld2View2
    out = {}
    rc = ref.read_intrinsics_binary(os.path.join(folder, "cameras.bin"))
    out["cam_ids"] = np.array(sorted(rc))
    for cid in rc:
        out[f"cam{cid}_model"] = np.array(rc[cid].model)
        out[f"cam{cid}_wh"] = np.array([rc[cid].width, rc[cid].height])
        out[f"cam{cid}_params"] = rc[cid].params
    ri = ref.read_extrinsics_binary(os.path.join(folder, "images.bin"))
    out["img_ids"] = np.array(list(ri))          # file order
    centres = []
    for iid, im in ri.items():
        out[f"img{iid}_qvec"], out[f"img{iid}_tvec"], out[f"img{iid}_cam"] = im.qvec, im.tvec, np.array(im.camera_id)
        out[f"img{iid}_name"], out[f"img{iid}_xys"], out[f"img{iid}_p3d"] = np.array(im.name), im.xys.reshape(-1, 2), im.point3D_ids
        R = np.transpose(ref.qvec2rotmat(im.qvec))
        out[f"img{iid}_R"] = R
        centres.append(np.linalg.inv(getWorld2View2(R, np.array(im.tvec)))[:3, 3:4])
    cc = np.hstack(centres)
    centre = np.mean(cc, axis=1, keepdims=True)
    out["norm_translate"] = -centre.flatten()
    out["norm_radius"] = np.array(np.max(np.linalg.norm(cc - centre, axis=0, keepdims=True)) * 1.1)
    x, c, e = ref.read_points3D_binary(os.path.join(folder, "points3D.bin"))
    out["pts_xyz"], out["pts_rgb"], out["pts_err"] = x, c, e
    np.savez_compressed(os.path.join(HERE, "colmap_small.npz"), **out)
    print("wrote colmap_small/ and colmap_small.npz")


def ref_losses():
    """(5) ref_losses.npz -- outputs of the reference's own loss functions (utils/loss_utils.py: l1_loss :24-25, ssim :30-70,
    plane_loss :72-78, depth_normal_loss :111-117, _get_img_grad_weight :119-131, tv_loss :536-557) on small random inputs:
    they pin gs2m_losses' PyTorch expressions and, through them, the fused HIP kernels of csrc/loss_ops.hip and the fused
    SSIM.  The module imports cv2 (absent here; used by _erode_cv only) and the CUDA-only gaussian_renderer (used by the
    multi-view terms only) at its top: it is loaded with EMPTY placeholder modules under those two names -- none of the
    functions called below touches either."""
    import types
    ref = _load_ref_loss_utils()
    g = torch.Generator().manual_seed(4321)
    H, W, P = 24, 32, 200
    out = {}
    img = torch.randn(3, H, W, generator=g) * 0.4 + 0.5
    gt = torch.rand(3, H, W, generator=g)
    normal = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    sobel = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    wm = torch.rand(1, H, W, generator=g)
    pred1, pred3 = torch.rand(1, H, W, generator=g), torch.rand(3, H, W, generator=g)
    raw_scale = torch.randn(P, 3, generator=g) - 3.0
    vis = torch.rand(P, generator=g) < 0.6
    out.update(img=img.numpy(), gt=gt.numpy(), normal=normal.numpy(), sobel=sobel.numpy(), wm=wm.numpy(), pred1=pred1.numpy(),
               pred3=pred3.numpy(), raw_scale=raw_scale.numpy(), vis=vis.numpy())
    rgb = img.clamp(0, 1)
    out["l1"] = ref.l1_loss(rgb, gt).numpy()
    out["ssim"] = ref.ssim(rgb.unsqueeze(0), gt.unsqueeze(0)).numpy()
    out["img_grad_weight"] = ref._get_img_grad_weight(gt).numpy()
    out["depth_normal"] = ref.depth_normal_loss(normal, sobel, gt).numpy()
    out["depth_normal_wm"] = ref.depth_normal_loss(normal, sobel, gt, weight_map=wm).numpy()
    out["tv_l2_c1"] = ref.tv_loss(gt, pred1, norm1=False).numpy()
    out["tv_l1_c3"] = ref.tv_loss(gt, pred3).numpy()
    out["tv_l1_c3_wm"] = ref.tv_loss(gt, pred3, weight_map=wm).numpy()
    out["plane"] = np.asarray(ref.plane_loss(vis, types.SimpleNamespace(get_scaling=torch.exp(raw_scale))))
    out["plane_none_visible"] = np.asarray(ref.plane_loss(torch.zeros(P, dtype=torch.bool), types.SimpleNamespace(get_scaling=torch.exp(raw_scale))), dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_losses.npz"), **out)
    print("wrote ref_losses.npz", {k: float(v) for k, v in out.items() if np.asarray(v).ndim == 0})


def _load_ref_loss_utils():
    """utils/loss_utils.py with EMPTY placeholder modules for its two top-level imports that cannot be satisfied here (cv2:
    absent, used by _erode_cv only; gaussian_renderer: CUDA-only, used by multi_view_loss / roughness_loss only)."""
    import importlib.util
    import types
    added = []
    for name in ("cv2", "gaussian_renderer"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.render = None
            sys.modules[name] = m
            added.append(name)
    spec = importlib.util.spec_from_file_location("ref_loss_utils", "/root/reference/utils/loss_utils.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for name in added:
        del sys.modules[name]
    return ref


def ref_mvs():
    """(6) ref_mvs.npz -- outputs of the reference's own pure helpers of the multi-view terms (utils/loss_utils.py:
    _patch_gradient :234-240, _sample_depth_normal :366-414, _sample_normal_map :432-453, _patch_offsets :455-457,
    _patch_warp :459-468, _loss_ncc :470-509): they pin gs2m_mvs' restatements, against which the fused patch-NCC and
    multi-view geometry kernels are tested.  (_get_points_from_depth and _reproject_points move tensors with .cuda() and
    cannot run here.)"""
    import types
    ref = _load_ref_loss_utils()
    g = torch.Generator().manual_seed(9876)
    out = {}
    for h in (1, 3):
        out[f"offsets_h{h}"] = ref._patch_offsets(h, "cpu").numpy()
    B, P = 13, 49
    Hm = torch.eye(3).repeat(B, 1, 1) + 0.05 * torch.randn(B, 3, 3, generator=g)
    uv = torch.rand(B, P, 2, generator=g) * 60.0
    out.update(warp_H=Hm.numpy(), warp_uv=uv.numpy(), warp_grid=ref._patch_warp(Hm.reshape(B, 9), uv).numpy())
    refp, neap = torch.rand(B, P, generator=g), torch.rand(B, P, generator=g)
    neap[:4] = refp[:4] * 0.7 + 0.1          # well correlated patches
    refp[4] = 0.5                            # a flat patch: the std mask
    ncc, mask = ref._loss_ncc(refp, neap)
    ncc2, smask = ref._loss_ncc(refp, neap, std_mask=True)
    out.update(ncc_ref=refp.numpy(), ncc_nea=neap.numpy(), ncc=ncc.numpy(), ncc_mask=mask.numpy(), ncc_std_mask=smask.numpy())
    out["patch_gradient"] = ref._patch_gradient(refp, 7).numpy()
    Hh, Ww = 20, 28
    normal_map = torch.nn.functional.normalize(torch.randn(3, Hh, Ww, generator=g), dim=0)
    depth_map = 2.0 + torch.rand(1, Hh, Ww, generator=g)
    ys, xs = torch.meshgrid(torch.arange(Hh, dtype=torch.float32), torch.arange(Ww, dtype=torch.float32), indexing="ij")
    pixels = torch.stack([xs, ys], dim=-1)
    out.update(sn_normal_map=normal_map.numpy(), sn_pixels=pixels.numpy(), sn_out=ref._sample_normal_map(pixels, normal_map).numpy())
    cam = types.SimpleNamespace(Fx=30.0, Fy=29.0, Cx=14.0, Cy=10.0, image_width=Ww, image_height=Hh)
    pts = torch.randn(300, 3, generator=g) * torch.tensor([1.2, 0.9, 1.0]) + torch.tensor([0.0, 0.0, 2.5])
    z, n, valid = ref._sample_depth_normal(pts, cam, {"depth_map": depth_map, "normal_map": normal_map})
    out.update(sdn_depth_map=depth_map.numpy(), sdn_pts=pts.numpy(), sdn_cam=np.array([cam.Fx, cam.Fy, cam.Cx, cam.Cy, Ww, Hh]),
               sdn_z=z.numpy(), sdn_n=n.numpy(), sdn_valid=valid.numpy())
    np.savez_compressed(os.path.join(HERE, "ref_mvs.npz"), **out)
    print("wrote ref_mvs.npz")


def _load_ref_pbr():
    """pbr/light.py and pbr/shade.py with EMPTY placeholder modules for the imports that cannot be satisfied here (cv2,
    nvdiffrast.torch, render_utils: absent / CUDA-only).  Only functions that never touch them are called below."""
    import importlib.util
    import types
    added = []
    for name in ("cv2", "nvdiffrast", "nvdiffrast.torch", "render_utils"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.diffuse_cubemap = m.specular_cubemap = None
            sys.modules[name] = m
            added.append(name)
    sys.modules["nvdiffrast"].torch = sys.modules["nvdiffrast.torch"]
    mods = {}
    pkg = types.ModuleType("ref_pbr"); pkg.__path__ = ["/root/reference/pbr"]; sys.modules["ref_pbr"] = pkg
    for name in ("light", "shade"):
        spec = importlib.util.spec_from_file_location("ref_pbr." + name, "/root/reference/pbr/%s.py" % name)
        mods[name] = importlib.util.module_from_spec(spec)
        sys.modules["ref_pbr." + name] = mods[name]
        spec.loader.exec_module(mods[name])
    for name in added + ["ref_pbr", "ref_pbr.light", "ref_pbr.shade"]:
        sys.modules.pop(name, None)
    return mods["light"], mods["shade"]


def ref_pbr():
    """(8) ref_pbr.npz -- outputs of the reference's own pure-PyTorch shading helpers (pbr/shade.py: saturate_dot :27,
    aces_film :32, linear_to_srgb :46, srgb_to_linear :61, rgb_to_srgb :95, srgb_to_rgb :112, envBRDF_approx :14; pbr/light.py:
    cube_to_dir :13, cubemap_mip.forward :31, CubemapLight.get_mip :75) and a 32 x 32 sub-sample of the environment-BRDF
    table it ships (pbr/brdf_256_256.bin, a data file), which tools/make_brdf_lut.py's table is held to."""
    import types
    light, shade = _load_ref_pbr()
    g = torch.Generator().manual_seed(4321)
    out = {}
    a = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    b = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    out["dot_a"], out["dot_b"] = a.numpy(), b.numpy()
    out["saturate_dot"] = shade.saturate_dot(a, b).numpy()
    x = torch.cat([torch.rand(500, generator=g) * 1.5 - 0.1, torch.tensor([0.0, 0.0031308, 0.003, 0.0032, 0.04045, 0.04, 0.041, 1.0])]).reshape(-1, 1, 4)[:, :, :3].contiguous()
    out["tone_x"] = x.numpy()
    out["aces_film"] = shade.aces_film(x).numpy()
    out["aces_film_np"] = shade.aces_film(x.numpy())
    out["linear_to_srgb"] = shade.linear_to_srgb(x).numpy()
    out["linear_to_srgb_np"] = shade.linear_to_srgb(x.numpy())
    out["srgb_to_linear"] = shade.srgb_to_linear(x).numpy()
    out["srgb_to_linear_np"] = shade.srgb_to_linear(x.numpy())
    x3 = x.reshape(1, 1, -1, 3).expand(1, 2, -1, 3).contiguous()
    out["rgb_to_srgb"] = shade.rgb_to_srgb(x3).numpy()
    out["srgb_to_rgb"] = shade.srgb_to_rgb(x3).numpy()
    rough = torch.rand(97, 1, generator=g); nov = torch.rand(97, 1, generator=g)
    out["env_rough"], out["env_nov"] = rough.numpy(), nov.numpy()
    out["envBRDF_approx"] = shade.envBRDF_approx(rough, nov).numpy()
    gx = torch.rand(5, 7, generator=g) * 2 - 1; gy = torch.rand(5, 7, generator=g) * 2 - 1
    out["cube_x"], out["cube_y"] = gx.numpy(), gy.numpy()
    out["cube_to_dir"] = np.stack([light.cube_to_dir(s, gx, gy).numpy() for s in range(6)])
    cm = torch.rand(6, 8, 8, 3, generator=g)
    out["mip_in"] = cm.numpy()
    out["mip_out"] = light.cubemap_mip.forward(None, cm).numpy()
    r = torch.cat([torch.rand(200, generator=g), torch.tensor([0.0, 0.04, 0.5, 0.4999, 1.0])]).reshape(-1, 1)
    out["mip_rough"] = r.numpy()
    for levels in (7, 4):
        fake = types.SimpleNamespace(MIN_ROUGHNESS=light.CubemapLight.MIN_ROUGHNESS, MAX_ROUGHNESS=light.CubemapLight.MAX_ROUGHNESS, specular=[None] * levels)
        out["get_mip_%d" % levels] = light.CubemapLight.get_mip(fake, r).numpy()
    out["light_consts"] = np.array([light.CubemapLight.LIGHT_MIN_RES, light.CubemapLight.MIN_ROUGHNESS, light.CubemapLight.MAX_ROUGHNESS])
    lut = np.fromfile("/root/reference/pbr/brdf_256_256.bin", dtype=np.float32).reshape(256, 256, 2)
    idx = np.arange(32) * 8 + 3
    out["brdf_idx"] = idx
    out["brdf_sub"] = lut[np.ix_(idx, idx)]
    np.savez_compressed(os.path.join(HERE, "ref_pbr.npz"), **out)
    print("wrote ref_pbr.npz")


def raster_small():
    import helpers as Hh
    from oracle import oracle
    sc = Hh.make_scene(400, 64, 48, seed=77, fc=10, scale_hi=0.08, bg=(0.1, 0.4, 0.7))
    f, gr = Hh.run_oracle(oracle, sc)
    cam = sc["cam"]
    out = dict(W=64, H=48, fc=10, sh_degree=3, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
               viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(), campos=cam["campos"].numpy(),
               bg=sc["bg"].numpy(), Gc=sc["Gc"].numpy().astype(np.float32), Gb=sc["Gb"].numpy().astype(np.float32),
               color=f.color, buffer=f.buffer, radii=f.radii, observe=f.observe, n_contrib=f.n_contrib,
               final_T=f.final_T, num_rendered=f.num_rendered, vals_sorted=f.vals_sorted, ranges=f.ranges)
    for k, v in sc["g"].items():
        out["in_" + k] = v.numpy()
    for k, v in gr.items():
        out["grad_" + k] = v
    np.savez_compressed(os.path.join(HERE, "raster_small.npz"), **out)
    print("wrote raster_small.npz", f.num_rendered)


if __name__ == "__main__":
    if "--model-only" in sys.argv:
        ref_model_helpers()
        sys.exit(0)
    if "--defaults-only" in sys.argv:
        ref_defaults()
        sys.exit(0)
    if "--colmap-only" in sys.argv:
        colmap_small()
        sys.exit(0)
    if "--losses-only" in sys.argv:
        ref_losses()
        ref_mvs()
        sys.exit(0)
    if "--pbr-only" in sys.argv:
        ref_pbr()
        sys.exit(0)
    ref_helpers()
    ref_model_helpers()
    ref_defaults()
    colmap_small()
    ref_losses()
    ref_mvs()
    ref_pbr()
    raster_small()
