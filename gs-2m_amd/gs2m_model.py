"""GaussianModel: the trainable state around the rasterizer -- parameters, optimizer groups,
densify / clone / split / prune with the Adam-state surgery, opacity resets, and the PLY checkpoint format.

Mirrors scene/gaussian_model.py (method names, argument meaning, tensor layouts, thresholds) so the reference's
train.py can drive it; written for PyTorch-ROCm with the fused optimizer (`gs2m_optim.Adam`, one launch for the nine
groups) and this repository's distCUDA2.  What is deliberately different:
  * the densification statistics use masked arithmetic instead of boolean-mask gathers and scatters (same values:
    adding 0 where the filter is off) -- no nonzero() / index_put_ sorts on the per-iteration path;
  * PLY I/O is a numpy structured-array reader / writer (plyfile is not a dependency): binary_little_endian 1.0, one
    `vertex` element, float32 properties in the reference's order (GM:260-278)
        x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3 albedo_0..2 roughness metallic
    with the SH tensors stored channel-major ((P, C, 3) -> transpose(1, 2) -> flatten), so files are interchangeable.
"""
import os

import numpy as np
import torch
from torch import nn

import gs2m_optim
from gs2m_scene import GaussianParams, build_rotation, inverse_sigmoid

SH_C0 = 0.28209479177387814


def rgb_to_sh(rgb):  # utils/sh_utils.py:114-115
    return (rgb - 0.5) / SH_C0


def expon_lr(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """utils/general_utils.py:36-66: log-linear interpolation with an optional delayed start."""
    def at(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return at


class OptimizationParams:
    """arguments/__init__.py:81-134 defaults that the model and the training loop read (pinned against the reference's class
    by tests/golden/ref_defaults.json)."""
    iterations = 30_000
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 30_000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.005
    rotation_lr = 0.001
    percent_dense = 0.001
    prune_init_points = True
    lambda_ssim = 0.2
    lambda_plane = 100.0
    lambda_depth_normal = 0.03
    lambda_multi_view = 1.0
    lambda_normal = 0.1
    lambda_smooth = 0.0
    lambda_rough = 1e-4
    lambda_alpha = 0.2
    geometry_from_iter = 5000
    material_from_iter = 30_000
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_from_iter = 500
    densify_until_iter = 15_000
    densify_grad_threshold = 0.0002
    densify_grad_abs_threshold = 0.0008
    opacity_prune_threshold = 0.005
    radii2D_threshold = 20
    use_opacity_reduce = False
    opacity_reduce_interval = 500
    use_multi_view_trim = True


_GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "albedo", "roughness", "metallic")
_ATTR = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
             rotation="_rotation", albedo="_albedo", roughness="_roughness", metallic="_metallic")


class GaussianModel(GaussianParams):
    def __init__(self, sh_degree=3, device="cuda"):
        self.device = device
        e = lambda *s: torch.empty(s, device=device)
        M = (sh_degree + 1) ** 2
        super().__init__(e(0, 3), e(0, 1, 3), e(0, M - 1, 3), e(0, 3), e(0, 4), e(0, 1), e(0, 3), e(0, 1), e(0, 1),
                         active_sh_degree=0, max_sh_degree=sh_degree)
        self.max_radii2D = e(0)
        self.xyz_gradient_accum = self.xyz_gradient_accum_abs = self.denom = e(0, 1)
        self.optimizer = None
        self.percent_dense = 0.0
        self.spatial_lr_scale = 0.0

    # ------------------------------------------------------------------ construction
    def parameterize(self, params):  # GM:205-222
        for name, t in zip(("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "albedo", "roughness", "metallic"), params):
            setattr(self, _ATTR[name], nn.Parameter(t.to(self.device).float().contiguous().requires_grad_(True)))

    def oneupSHdegree(self):  # GM:174-176
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def create_from_pcd(self, points, colors, spatial_lr_scale):
        """GM:178-203.  points, colors: (n, 3) arrays / tensors (colours in [0, 1])."""
        from simple_knn._C import distCUDA2
        self.spatial_lr_scale = spatial_lr_scale
        xyz = torch.as_tensor(np.asarray(points)).float().to(self.device)
        n = xyz.shape[0]
        M = (self.max_sh_degree + 1) ** 2
        feats = torch.zeros((n, M, 3), device=self.device)
        feats[:, 0, :] = rgb_to_sh(torch.as_tensor(np.asarray(colors)).float().to(self.device))
        dist2 = torch.clamp_min(distCUDA2(xyz), 1e-7)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=self.device)
        rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones((n, 1), device=self.device))
        ones = lambda c: torch.ones((n, c), device=self.device)
        self.parameterize((xyz, feats[:, :1].contiguous(), feats[:, 1:].contiguous(), scales, rots, opac, ones(3), ones(1), ones(1)))
        self.max_radii2D = torch.zeros(n, device=self.device)

    def training_setup(self, args=OptimizationParams, optimizer_cls=None):  # GM:224-249
        self.percent_dense = args.percent_dense
        self._reset_stats()
        if self.max_radii2D.shape[0] != self._xyz.shape[0]:  # a model that came from parameterize() / load_ply()
            self.max_radii2D = torch.zeros(self._xyz.shape[0], device=self.device)
        lr = dict(xyz=args.position_lr_init * self.spatial_lr_scale, f_dc=args.feature_lr, f_rest=args.feature_lr / 20.0,
                  opacity=args.opacity_lr, scaling=args.scaling_lr, rotation=args.rotation_lr, albedo=args.opacity_lr,
                  roughness=args.opacity_lr, metallic=args.opacity_lr)
        groups = [{"params": [getattr(self, _ATTR[n])], "lr": lr[n], "name": n} for n in _GROUPS]
        self.optimizer = (optimizer_cls or gs2m_optim.Adam)(groups, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = expon_lr(args.position_lr_init * self.spatial_lr_scale, args.position_lr_final * self.spatial_lr_scale,
                                           lr_delay_mult=args.position_lr_delay_mult, max_steps=args.position_lr_max_steps)
        if args.prune_init_points:
            self.prune_init_points()

    def update_learning_rate(self, iteration):  # GM:251-258
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)
                return group["lr"]

    def _reset_stats(self):
        n = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.xyz_gradient_accum_abs = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)

    # ------------------------------------------------------------------ PLY (GM:260-360)
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
        return names + [f"albedo_{i}" for i in range(3)] + ["roughness", "metallic"]

    def save_ply(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        c = lambda t: t.detach().cpu().numpy()
        xyz = c(self._xyz)
        cols = [xyz, np.zeros_like(xyz), c(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1)),
                c(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1)), c(self._opacity), c(self._scaling),
                c(self._rotation), c(self._albedo), c(self._roughness), c(self._metallic)]
        table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
        names = self.construct_list_of_attributes()
        assert table.shape[1] == len(names)
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
        header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(table.tobytes())

    @staticmethod
    def read_ply_vertices(path):
        """-> {property name: (n,) float array}.  binary_little_endian / ascii PLY with one scalar-property vertex element."""
        np_types = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1",
                    "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4",
                    "uint": "<u4", "uint32": "<u4"}
        with open(path, "rb") as f:
            if f.readline().strip() != b"ply":
                raise ValueError(f"{path}: not a PLY file")
            fmt, n, props, in_vertex = None, None, [], False
            while True:
                line = f.readline()
                if not line:
                    raise ValueError(f"{path}: truncated PLY header")
                tok = line.decode("ascii").split()
                if not tok or tok[0] == "comment":
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    in_vertex = tok[1] == "vertex"
                    if in_vertex:
                        n = int(tok[2])
                    elif n is None:
                        raise ValueError(f"{path}: the vertex element must come first")
                elif tok[0] == "property" and in_vertex:
                    if tok[1] == "list":
                        raise ValueError(f"{path}: list properties in the vertex element are not supported")
                    props.append((tok[2], np_types[tok[1]]))
                elif tok[0] == "end_header":
                    break
            if n is None:
                raise ValueError(f"{path}: no vertex element")
            if fmt == "binary_little_endian":
                data = np.frombuffer(f.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
                return {name: np.asarray(data[name]) for name, _ in props}
            if fmt == "ascii":
                rows = np.loadtxt(f, max_rows=n, ndmin=2)
                return {name: rows[:, i] for i, (name, _) in enumerate(props)}
            raise ValueError(f"{path}: unsupported PLY format {fmt}")

    def load_ply(self, path):
        v = self.read_ply_vertices(path)
        col = lambda *names: np.stack([np.asarray(v[n], dtype=np.float32) for n in names], axis=1)
        order = lambda prefix: sorted((n for n in v if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
        rest = order("f_rest_")
        M = (self.max_sh_degree + 1) ** 2
        assert len(rest) == 3 * M - 3, f"{path}: {len(rest)} f_rest properties, SH degree {self.max_sh_degree} needs {3 * M - 3}"
        t = lambda a: torch.tensor(a, dtype=torch.float, device=self.device)
        f_dc = t(col("f_dc_0", "f_dc_1", "f_dc_2")).reshape(-1, 3, 1).transpose(1, 2)
        f_rest = t(col(*rest)).reshape(-1, 3, M - 1).transpose(1, 2)
        self.parameterize((t(col("x", "y", "z")), f_dc, f_rest, t(col(*order("scale_"))), t(col(*order("rot"))), t(col("opacity")),
                           t(col("albedo_0", "albedo_1", "albedo_2")), t(col("roughness")), t(col("metallic"))))
        self.active_sh_degree = self.max_sh_degree
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=self.device)

    # ------------------------------------------------------------------ optimizer surgery (GM:372-455)
    def _rebind(self, tensors):
        for name, p in tensors.items():
            setattr(self, _ATTR[name], p)

    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] != name:
                continue
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(tensor.requires_grad_(True))
            if state is not None:
                state["exp_avg"], state["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
                self.optimizer.state[new] = state
            group["params"][0] = new
            out[name] = new
        return out

    def _edit_groups(self, edit):
        """edit(name, tensor) -> tensor, applied to every parameter and its two Adam moments."""
        out = {}
        for group in self.optimizer.param_groups:
            assert len(group["params"]) == 1
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(edit(group["name"], old.data, False).contiguous().requires_grad_(True))
            if state is not None:
                state["exp_avg"] = edit(group["name"], state["exp_avg"], True).contiguous()
                state["exp_avg_sq"] = edit(group["name"], state["exp_avg_sq"], True).contiguous()
                self.optimizer.state[new] = state
            group["params"][0] = new
            out[group["name"]] = new
        return out

    def _prune_optimizer(self, mask):
        return self._edit_groups(lambda name, t, is_state: t[mask])

    def cat_tensors_to_optimizer(self, tensors_dict):
        return self._edit_groups(lambda name, t, is_state: torch.cat(
            (t, torch.zeros_like(tensors_dict[name]) if is_state else tensors_dict[name]), dim=0))

    def prune_points(self, mask):
        keep = ~mask
        self._rebind(self._prune_optimizer(keep))
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.xyz_gradient_accum_abs = self.xyz_gradient_accum_abs[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def prune_init_points(self):  # GM:417-426
        big = torch.max(self.get_scaling, dim=1).values
        m1 = big > torch.mean(self.get_scaling)
        if len(self.get_scaling) < 500_0000:
            m2 = big > torch.quantile(self.get_scaling, 0.999)
        else:
            m2 = big > torch.mean(self.get_scaling) * 4
        self.prune_points(torch.logical_and(m1, m2))

    def densification_postfix(self, **new):
        self._rebind(self.cat_tensors_to_optimizer(new))
        self._reset_stats()
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=self.device)

    # ------------------------------------------------------------------ densification (GM:489-573)
    def _selected(self, sel, repeat=1):
        rep = lambda t: t[sel].repeat(repeat, *([1] * (t.dim() - 1)))
        return dict(f_dc=rep(self._features_dc), f_rest=rep(self._features_rest), opacity=rep(self._opacity),
                    rotation=rep(self._rotation), albedo=rep(self._albedo), roughness=rep(self._roughness), metallic=rep(self._metallic))

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        n = self._xyz.shape[0]
        padded = torch.zeros(n, device=self.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = torch.logical_and(padded >= grad_threshold, torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new = self._selected(sel, N)
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self._xyz[sel].repeat(N, 1)
        new["scaling"] = torch.log(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(**new)
        self.prune_points(torch.cat((sel, torch.zeros(N * int(sel.sum()), device=self.device, dtype=torch.bool))))

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = torch.logical_and(torch.norm(grads, dim=-1) >= grad_threshold,
                                torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        new = self._selected(sel)
        new["xyz"], new["scaling"] = self._xyz[sel], self._scaling[sel]
        self.densification_postfix(**new)

    def densify_and_prune(self, max_grad, max_grad_abs, min_opacity, extent, max_screen_size=None):
        grads = torch.nan_to_num(self.xyz_gradient_accum / self.denom, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        grads_abs = torch.nan_to_num(self.xyz_gradient_accum_abs / self.denom, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads_abs, max_grad_abs, extent)
        prune = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)

    def add_densification_stats(self, viewspace_points, update_filter):
        """GM:569-573 in masked form: rows outside the filter add exactly 0."""
        g = viewspace_points.grad
        f = update_filter[:, None]
        self.xyz_gradient_accum += torch.where(f, torch.norm(g[:, :2], dim=-1, keepdim=True), 0.0)
        self.xyz_gradient_accum_abs += torch.where(f, torch.norm(g[:, 2:], dim=-1, keepdim=True), 0.0)
        self.denom += f

    def update_max_radii(self, observe, visibility_filter, radii):  # train.py:223-225
        mask = (observe > 0) & visibility_filter
        self.max_radii2D = torch.where(mask, torch.max(self.max_radii2D, radii), self.max_radii2D)

    def accumulate_view_stats(self, viewspace_points, visibility_filter, observe, radii, fused=None):
        """train.py:223-227 for one view: `update_max_radii` + `add_densification_stats`.  On the GPU one HIP launch
        (gs2m_losses.densification_stats) instead of ~14 framework kernels; `fused=False` keeps the PyTorch expressions."""
        if fused is None:
            fused = self._xyz.is_cuda
        if fused and observe.dtype == torch.int32 and radii.dtype == torch.int32:
            from gs2m_losses import densification_stats
            densification_stats(viewspace_points.grad, visibility_filter, self.xyz_gradient_accum, self.xyz_gradient_accum_abs, self.denom,
                                observe, radii, self.max_radii2D)
        else:
            self.update_max_radii(observe, visibility_filter, radii)
            self.add_densification_stats(viewspace_points, visibility_filter)

    def reset_opacity(self, ceiling=0.01):  # GM:362-365 (ceiling 0.8: reduce_opacity, GM:367-370)
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * ceiling))
        self._rebind(self.replace_tensor_to_optimizer(new, "opacity"))

    def reduce_opacity(self):
        self.reset_opacity(0.8)
