"""Dense PyTorch-autograd restatement of the rasterizer (float64, small scenes only).

Secondary checker for the C oracle: it derives gradients by
autograd instead of the hand-written backward, so it only agrees with the reference
semantics after the reference's non-autograd quirks are patched in explicitly:
  Q1  alpha = min(0.99, o*G) is straight-through in the backward (CR/backward.cu:519-533)
  Q2  d(conic)/d(cov2D) is evaluated at cov2D + 0.3*I although the forward never adds it
      (CR/backward.cu:205-207 vs CR/forward.cu:101-103)
  Q3  the frustum-clamped t.x, t.y carry no gradient at all when clamped
      (CR/backward.cu:183-184, 268-270)
Depth keys, radii and tile rectangles are non-differentiable in both.
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh_color(deg, sh, dirs):
    """sh (P,M,3), dirs (P,3) unit -> (P,3) before +0.5/clamp (CR/forward.cu:20-67)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14] + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res


class _ConicQ2(torch.autograd.Function):
    """conic = inverse(cov2D) forward; backward = VJP of the inverse evaluated at cov2D + 0.3 I."""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b, c)
        det = a * c - b * b
        return c / det, -b / det, a / det

    @staticmethod
    def backward(ctx, gx, gy, gz):
        a, b, c = ctx.saved_tensors
        a = a + 0.3
        c = c + 0.3
        det = a * c - b * b
        d2 = 1.0 / (det * det + 1e-7)
        # true derivatives of (c/det, -b/det, a/det) w.r.t. (a, b, c)
        da = d2 * (-c * c * gx + b * c * gy + (det - a * c) * gz)
        dc = d2 * (-a * a * gz + a * b * gy + (det - a * c) * gx)
        db = d2 * (2 * b * c * gx - (det + 2 * b * b) * gy + 2 * a * b * gz)
        return da, db, dc


def rasterize_dense(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, features, *,
                    bg, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree, scale_modifier=1.0,
                    feature_count=0, tiled=False):
    """Returns (color (3,H,W), buffer (10,H,W), radii (P), aux).  All float64 tensors.
    tiled=True composites tile by tile over the Gaussians whose rectangle reaches the tile (same per-pixel sequence:
    a Gaussian outside a pixel's tile rectangle never passes `in_rect`), which keeps the autograd graph proportional
    to the number of (tile, Gaussian) instances instead of pixels x Gaussians -- config C1 (10k Gaussians, 256x256)
    then fits in a few seconds and ~1 GB."""
    dt = torch.float64
    P = means3D.shape[0]
    vm = viewmatrix.to(dt)
    pm = projmatrix.to(dt)
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ vm  # row-vector convention == column-major W2C
    p_hom = hom @ pm
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    visible = p_view[:, 2] > 0.2

    if cov3D_precomp is None:
        s = scale_modifier * scales
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rm = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        Sigma = Rm @ torch.diag_embed(s * s) @ Rm.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(P, 3, 3)

    t = p_view[:, :3]
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), t[:, 0])  # Q3
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(P, 2, 3)
    Rw2c = vm[:3, :3].transpose(0, 1)  # W2C rotation (vm holds its transpose)
    JR = J @ Rw2c
    cov2 = JR @ Sigma @ JR.transpose(1, 2)
    a, b, c_ = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det = a * c_ - b * b
    ok = visible & (det != 0)
    det_safe = torch.where(ok, det, torch.ones_like(det))
    a_s = torch.where(ok, a, torch.ones_like(a))
    b_s = torch.where(ok, b, torch.zeros_like(b))
    c_s = torch.where(ok, c_, torch.ones_like(c_))
    A, B, Cc = _ConicQ2.apply(a_s, b_s, c_s)
    with torch.no_grad():
        mid = 0.5 * (a_s + c_s)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det_safe, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(lam))
        gx_, gy_ = (W + 15) // 16, (H + 15) // 16
        rminx = torch.clamp(torch.trunc((pix[:, 0] - radius) / 16), 0, gx_)
        rminy = torch.clamp(torch.trunc((pix[:, 1] - radius) / 16), 0, gy_)
        rmaxx = torch.clamp(torch.trunc((pix[:, 0] + radius + 15) / 16), 0, gx_)
        rmaxy = torch.clamp(torch.trunc((pix[:, 1] + radius + 15) / 16), 0, gy_)
        ok = ok & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
        radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh_color(sh_degree, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp

    with torch.no_grad():
        order = torch.argsort(p_view[:, 2].float(), stable=True)  # fp32 depth key, ties by index
        order = order[ok[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pxf_all, pyf_all = xs.reshape(-1), ys.reshape(-1)
    N = W * H
    bgd = bg.to(dt)

    def composite(cand, pxf, pyf, tile_x, tile_y):
        """front-to-back over the Gaussians `cand` (depth order) on the given pixels -> (color (3,n), buf (10,n), T (n))"""
        n = pxf.shape[0]
        color = torch.zeros(3, n, dtype=dt)
        buf = [torch.zeros(n, dtype=dt) for _ in range(10)]
        T = torch.ones(n, dtype=dt)
        done = torch.zeros(n, dtype=torch.bool)
        for i in cand:
            with torch.no_grad():
                in_rect = (tile_x >= rminx[i]) & (tile_x < rmaxx[i]) & (tile_y >= rminy[i]) & (tile_y < rmaxy[i])
            dx = pix[i, 0] - pxf
            dy = pix[i, 1] - pyf
            power = -0.5 * (A[i] * dx * dx + Cc[i] * dy * dy) - B[i] * dx * dy
            Gv = torch.exp(power)
            raw = opacities[i, 0] * Gv
            alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # Q1
            with torch.no_grad():
                m = in_rect & (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
                test_T = T * (1 - alpha)
                term = m & (test_T < 1e-4)
                done = done | term
                m = m & (~term)
            w = torch.where(m, alpha * T, torch.zeros_like(T))
            color = color + rgb[i][:, None] * w[None]
            for ch in range(feature_count):
                buf[ch] = buf[ch] + features[i, ch] * w
            T = torch.where(m, T * (1 - alpha), T)
        color = color + T[None] * bgd[:, None]
        return color, torch.stack(buf, 0), T

    if not tiled:
        color, buffer, T = composite([int(i) for i in order], pxf_all, pyf_all, torch.floor(pxf_all / 16), torch.floor(pyf_all / 16))
        return color.reshape(3, H, W), buffer.reshape(10, H, W), radii, dict(final_T=T.reshape(H, W).detach())
    gx_, gy_ = (W + 15) // 16, (H + 15) // 16
    color = torch.zeros(3, H, W, dtype=dt)
    buffer = torch.zeros(10, H, W, dtype=dt)
    final_T = torch.zeros(H, W, dtype=dt)
    rx0, rx1, ry0, ry1 = rminx[order], rmaxx[order], rminy[order], rmaxy[order]
    pieces = []
    for ty in range(gy_):
        for tx in range(gx_):
            with torch.no_grad():
                cand = order[(rx0 <= tx) & (tx < rx1) & (ry0 <= ty) & (ty < ry1)]
            y0, y1, x0, x1 = ty * 16, min(ty * 16 + 16, H), tx * 16, min(tx * 16 + 16, W)
            pyf, pxf = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
            pxf, pyf = pxf.reshape(-1), pyf.reshape(-1)
            c, b_, T = composite([int(i) for i in cand], pxf, pyf, torch.full_like(pxf, float(tx)), torch.full_like(pyf, float(ty)))
            pieces.append((y0, y1, x0, x1, c, b_, T))
    # assemble without in-place writes into a tensor that requires grad: rows of tiles are concatenated
    rows_c, rows_b = [], []
    k = 0
    for ty in range(gy_):
        rc, rb = [], []
        for tx in range(gx_):
            y0, y1, x0, x1, c, b_, T = pieces[k]; k += 1
            rc.append(c.reshape(3, y1 - y0, x1 - x0)); rb.append(b_.reshape(10, y1 - y0, x1 - x0))
            final_T[y0:y1, x0:x1] = T.detach().reshape(y1 - y0, x1 - x0)
        rows_c.append(torch.cat(rc, dim=2)); rows_b.append(torch.cat(rb, dim=2))
    color, buffer = torch.cat(rows_c, dim=1), torch.cat(rows_b, dim=1)
    return color, buffer, radii, dict(final_T=final_T)
