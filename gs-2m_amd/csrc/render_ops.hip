// Fused pre- and post-processing of render() around the rasterizer, for gfx950.
//
// The reference does this work with ~25 small PyTorch ops per view (gaussian_renderer/__init__.py:83-96,
// 126-141; scene/gaussian_model.py:146-160), among them three (N,3)x(3,3) matmuls and a bmm.  On ROCm those
// K = 3 products go to the BLAS library and cost 22 ms per view at 1M Gaussians / 1080p -- ten times the
// rasterizer.  Everything here is elementwise per Gaussian or per pixel and HBM bound:
//   pack_features   xyz, scales, rotations, albedo, roughness, metallic -> the (P, 10) feature rows
//                   [1, distance, normal(3), albedo(3), roughness, metallic]  (normal: min-scale axis of the
//                   rotation, flipped to face the camera, GM:146-160; distance: |n_cam . p_cam| or z, GR:89)
//   gbuffer_post    blended G-buffer -> normal mask, camera-space normals, plane-distance -> depth (GR:126-141)
// plus their backward passes (the reference gets those from autograd).  Index/sign decisions (argmin of the
// scales, the flip, abs) are evaluated exactly as the PyTorch ops do.
#include "common.h"

namespace {

struct Mat3 {  // V[:3, :3] of the row-vector convention: out_j = sum_i in_i * m[i][j]
    float m[3][3];
    float t[3];  // V[3, :3]
};

// world_view_transform, row-major 4x4 in device memory (wave-uniform: scalar loads)
__device__ __forceinline__ Mat3 load_view(const float* __restrict__ v) {
    Mat3 V;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) V.m[i][j] = v[4 * i + j];
#pragma unroll
    for (int j = 0; j < 3; j++) V.t[j] = v[12 + j];
    return V;
}

__device__ __forceinline__ void quat_to_rot(const float q[4], float R[3][3]) {  // utils/general_utils.py:75-98
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// everything the forward and the backward share for one Gaussian
struct NormalEval {
    int k;           // min-scale axis
    float qn;        // |rot|
    float q[4];      // rot / |rot|
    float sgn;       // +-1: flip to face the camera
    float mlen;      // |column k of R|
    float n[3];      // unit normal, world space
};
__device__ __forceinline__ NormalEval eval_normal(const float* __restrict__ xyz, const float* __restrict__ scales,
                                                  const float* __restrict__ rot, const float* __restrict__ campos, int i) {
    NormalEval e;
    const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
    e.k = 0;  // torch.argmin: the first index of the minimum
    float sm = s0;
    if (s1 < sm) { sm = s1; e.k = 1; }
    if (s2 < sm) { e.k = 2; }
    const float4 rq = reinterpret_cast<const float4*>(rot)[i];
    e.qn = sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
    e.q[0] = rq.x / e.qn; e.q[1] = rq.y / e.qn; e.q[2] = rq.z / e.qn; e.q[3] = rq.w / e.qn;
    float R[3][3];
    quat_to_rot(e.q, R);
    float m[3] = {R[0][e.k], R[1][e.k], R[2][e.k]};
    const float vx = campos[0] - xyz[3 * i], vy = campos[1] - xyz[3 * i + 1], vz = campos[2] - xyz[3 * i + 2];
    e.sgn = (m[0] * vx + m[1] * vy + m[2] * vz) < 0.0f ? -1.f : 1.f;
    m[0] *= e.sgn; m[1] *= e.sgn; m[2] *= e.sgn;
    e.mlen = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    e.n[0] = m[0] / e.mlen; e.n[1] = m[1] / e.mlen; e.n[2] = m[2] / e.mlen;
    return e;
}

__global__ void __launch_bounds__(256) pack_features_kernel(int P, const float* __restrict__ xyz,
                                                            const float* __restrict__ scales,
                                                            const float* __restrict__ rot,
                                                            const float* __restrict__ albedo,
                                                            const float* __restrict__ roughness,
                                                            const float* __restrict__ metallic,
                                                            const float* __restrict__ campos, const float* __restrict__ view,
                                                            int z_depth, int blend_metallic,
                                                            float* __restrict__ features) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const Mat3 V = load_view(view);
    const NormalEval e = eval_normal(xyz, scales, rot, campos, i);
    const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    float cn[3], cp[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        cn[j] = e.n[0] * V.m[0][j] + e.n[1] * V.m[1][j] + e.n[2] * V.m[2][j];
        cp[j] = px * V.m[0][j] + py * V.m[1][j] + pz * V.m[2][j] + V.t[j];
    }
    const float dist = z_depth ? cp[2] : fabsf(cn[0] * cp[0] + cn[1] * cp[1] + cn[2] * cp[2]);
    float2* f2 = reinterpret_cast<float2*>(features + (size_t)i * GS2M_NUM_FEATURES);
    f2[0] = make_float2(1.0f, dist);
    f2[1] = make_float2(e.n[0], e.n[1]);
    f2[2] = make_float2(e.n[2], albedo[3 * i]);
    f2[3] = make_float2(albedo[3 * i + 1], albedo[3 * i + 2]);
    f2[4] = make_float2(roughness[i], blend_metallic ? metallic[i] : 0.0f);
}

__global__ void __launch_bounds__(256) pack_features_bwd_kernel(
    int P, const float* __restrict__ xyz, const float* __restrict__ scales, const float* __restrict__ rot,
    const float* __restrict__ campos, const float* __restrict__ view, int z_depth, int blend_metallic,
    const float* __restrict__ dF,
    float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_albedo,
    float* __restrict__ d_roughness, float* __restrict__ d_metallic) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const Mat3 V = load_view(view);
    const NormalEval e = eval_normal(xyz, scales, rot, campos, i);
    const float2* g2 = reinterpret_cast<const float2*>(dF + (size_t)i * GS2M_NUM_FEATURES);
    const float2 g0 = g2[0], g1 = g2[1], g2_ = g2[2], g3 = g2[3], g4 = g2[4];
    d_albedo[3 * i] = g2_.y; d_albedo[3 * i + 1] = g3.x; d_albedo[3 * i + 2] = g3.y;
    d_roughness[i] = g4.x;
    d_metallic[i] = blend_metallic ? g4.y : 0.0f;
    float dn[3] = {g1.x, g1.y, g2_.x};
    float dp[3] = {0.f, 0.f, 0.f};
    const float ddist = g0.y;
    if (z_depth) {
#pragma unroll
        for (int a = 0; a < 3; a++) dp[a] = ddist * V.m[a][2];
    } else {
        const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
        float cn[3], cp[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cn[j] = e.n[0] * V.m[0][j] + e.n[1] * V.m[1][j] + e.n[2] * V.m[2][j];
            cp[j] = px * V.m[0][j] + py * V.m[1][j] + pz * V.m[2][j] + V.t[j];
        }
        const float s = cn[0] * cp[0] + cn[1] * cp[1] + cn[2] * cp[2];
        const float ds = ddist * (s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f));  // abs backward: sign, 0 at 0
#pragma unroll
        for (int a = 0; a < 3; a++) {
            dn[a] += ds * (cp[0] * V.m[a][0] + cp[1] * V.m[a][1] + cp[2] * V.m[a][2]);
            dp[a] = ds * (cn[0] * V.m[a][0] + cn[1] * V.m[a][1] + cn[2] * V.m[a][2]);
        }
    }
    d_xyz[3 * i] = dp[0]; d_xyz[3 * i + 1] = dp[1]; d_xyz[3 * i + 2] = dp[2];
    // n = m / |m|,  m = sgn * R[:, k]
    const float ndn = e.n[0] * dn[0] + e.n[1] * dn[1] + e.n[2] * dn[2];
    float dc[3];  // gradient of column k of R
#pragma unroll
    for (int a = 0; a < 3; a++) dc[a] = e.sgn * (dn[a] - e.n[a] * ndn) / e.mlen;
    // column k of R as a function of q = (r, x, y, z)
    const float r = e.q[0], x = e.q[1], y = e.q[2], z = e.q[3];
    float dq[4];
    if (e.k == 0) {  // (1 - 2(y^2 + z^2), 2(xy + rz), 2(xz - ry))
        dq[0] = 2.f * (z * dc[1] - y * dc[2]);
        dq[1] = 2.f * (y * dc[1] + z * dc[2]);
        dq[2] = -4.f * y * dc[0] + 2.f * x * dc[1] - 2.f * r * dc[2];
        dq[3] = -4.f * z * dc[0] + 2.f * r * dc[1] + 2.f * x * dc[2];
    } else if (e.k == 1) {  // (2(xy - rz), 1 - 2(x^2 + z^2), 2(yz + rx))
        dq[0] = 2.f * (-z * dc[0] + x * dc[2]);
        dq[1] = 2.f * y * dc[0] - 4.f * x * dc[1] + 2.f * r * dc[2];
        dq[2] = 2.f * (x * dc[0] + z * dc[2]);
        dq[3] = -2.f * r * dc[0] - 4.f * z * dc[1] + 2.f * y * dc[2];
    } else {  // (2(xz + ry), 2(yz - rx), 1 - 2(x^2 + y^2))
        dq[0] = 2.f * (y * dc[0] - x * dc[1]);
        dq[1] = 2.f * z * dc[0] - 2.f * r * dc[1] - 4.f * x * dc[2];
        dq[2] = 2.f * r * dc[0] + 2.f * z * dc[1] - 4.f * y * dc[2];
        dq[3] = 2.f * (x * dc[0] + y * dc[1]);
    }
    // q = rot / |rot|
    const float qdq = e.q[0] * dq[0] + e.q[1] * dq[1] + e.q[2] * dq[2] + e.q[3] * dq[3];
    reinterpret_cast<float4*>(d_rot)[i] = make_float4((dq[0] - e.q[0] * qdq) / e.qn, (dq[1] - e.q[1] * qdq) / e.qn,
                                                      (dq[2] - e.q[2] * qdq) / e.qn, (dq[3] - e.q[3] * qdq) / e.qn);
}

// upstream gradients of the ten channels taken as plain slices of the buffer (NULL = none); all = 0: channels 1..4
// only, the caller owns the rest of d_buffer
struct GBufDirect {
    const float* g[10];
    int all;
};

__global__ void __launch_bounds__(256) gbuffer_post_kernel(int N, const float* __restrict__ buffer,
                                                           const float* __restrict__ rays, const float* __restrict__ view,
                                                           int z_depth, uint8_t* __restrict__ normal_mask,
                                                           float* __restrict__ local_normal,
                                                           float* __restrict__ depth) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const Mat3 V = load_view(view);
    const size_t n = (size_t)N;
    const float dist = buffer[n + p], a = buffer[2 * n + p], b = buffer[3 * n + p], c = buffer[4 * n + p];
    normal_mask[p] = (a != 0.f && b != 0.f && c != 0.f) ? 1 : 0;
    float ln[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        ln[j] = a * V.m[0][j] + b * V.m[1][j] + c * V.m[2][j];
        local_normal[j * n + p] = ln[j];
    }
    if (z_depth) {
        depth[p] = dist;
    } else {
        const float denom = ln[0] * rays[3 * (size_t)p] + ln[1] * rays[3 * (size_t)p + 1] + ln[2] * rays[3 * (size_t)p + 2];
        depth[p] = dist / -(denom + 1e-8f);
    }
}

// gradient with respect to buffer channels 1..4 (written into a (10, H, W) tensor whose other channels the caller
// zero-fills or already holds)
__global__ void __launch_bounds__(256) gbuffer_post_bwd_kernel(int N, const float* __restrict__ buffer,
                                                               const float* __restrict__ rays, const float* __restrict__ view,
                                                               int z_depth, const float* __restrict__ d_local_normal,
                                                               const float* __restrict__ d_depth, GBufDirect dir,
                                                               float* __restrict__ d_buffer) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const Mat3 V = load_view(view);
    const size_t n = (size_t)N;
    float dl[3] = {0.f, 0.f, 0.f};
    if (d_local_normal != nullptr) { dl[0] = d_local_normal[p]; dl[1] = d_local_normal[n + p]; dl[2] = d_local_normal[2 * n + p]; }
    const float dd = d_depth != nullptr ? d_depth[p] : 0.f;
    float ddist;
    if (z_depth) {
        ddist = dd;
    } else {
        const float dist = buffer[n + p], a = buffer[2 * n + p], b = buffer[3 * n + p], c = buffer[4 * n + p];
        const float r0 = rays[3 * (size_t)p], r1 = rays[3 * (size_t)p + 1], r2 = rays[3 * (size_t)p + 2];
        float denom = 0.f;
        const float rr[3] = {r0, r1, r2};
#pragma unroll
        for (int j = 0; j < 3; j++) denom += (a * V.m[0][j] + b * V.m[1][j] + c * V.m[2][j]) * rr[j];
        const float e = denom + 1e-8f;
        ddist = -dd / e;             // depth = -dist / e
        const float de = dd * dist / (e * e);
#pragma unroll
        for (int j = 0; j < 3; j++) dl[j] += de * rr[j];
    }
    // plus the gradients of the maps that are plain channel slices (alpha, distance, normal, albedo, roughness,
    // metallic): every channel of d_buffer is written here, once
    auto direct = [&](int c) -> float { return dir.g[c] != nullptr ? dir.g[c][p] : 0.f; };
    if (dir.all) {
        d_buffer[p] = direct(0);
#pragma unroll
        for (int c = 5; c < 10; c++) d_buffer[c * n + p] = direct(c);
    }
    d_buffer[n + p] = ddist + (dir.all ? direct(1) : 0.f);
#pragma unroll
    for (int a_ = 0; a_ < 3; a_++)
        d_buffer[(2 + a_) * n + p] = dl[0] * V.m[a_][0] + dl[1] * V.m[a_][1] + dl[2] * V.m[a_][2] + (dir.all ? direct(2 + a_) : 0.f);
}

// ---- normal from the depth map (GR:167-175 render_normal_from_depth_map; utils/normal_utils.py:3-72) --------------
// World point of pixel (u, v):  X = d * ray(u, v) + c,  ray = R_c2w K^-1 (u, v, 1),  c = camera centre;  normal =
// normalize((X_right - X_left) x (X_top - X_bottom)) at interior pixels, 0 on the border;  output = normal * alpha +
// background * (1 - alpha).  The reference builds this from ~40 PyTorch ops with two GEMMs over H*W rows and two
// torch.inverse calls; here the inverse of the 3x3 rotation is the adjugate (wave-uniform), K^-1 is closed form.
struct CamInv {
    float R[3][3];  // camera -> world rotation, column-vector convention: x_w = R x_c + c
    float fx, fy, cx, cy;
};
__device__ __forceinline__ CamInv load_cam_inv(const float* __restrict__ v, float fx, float fy, float cx, float cy) {
    // world_view_transform v (row-major 4x4, row-vector convention): W2C rotation A[r][c] = v[4c + r]
    float A[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) A[r][c] = v[4 * c + r];
    const float c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2],
                c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const float det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02, id = 1.0f / det;
    CamInv C;
    C.R[0][0] = c00 * id; C.R[1][0] = c01 * id; C.R[2][0] = c02 * id;
    C.R[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
    C.R[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id;
    C.R[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
    C.R[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
    C.R[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
    C.R[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
    C.fx = fx; C.fy = fy; C.cx = cx; C.cy = cy;
    return C;
}
__device__ __forceinline__ void world_ray(const CamInv& C, int u, int v, float r[3]) {
    const float x = ((float)u - C.cx) / C.fx, y = ((float)v - C.cy) / C.fy;
#pragma unroll
    for (int a = 0; a < 3; a++) r[a] = C.R[a][0] * x + C.R[a][1] * y + C.R[a][2];
}
struct Stencil {
    float a[3], b[3], n[3], len;  // a = X_right - X_left, b = X_top - X_bottom, n = normalize(a x b), len = |a x b|
};
__device__ __forceinline__ Stencil eval_stencil(const CamInv& C, const float* __restrict__ depth, int W, int u, int v) {
    float rr[3], rl[3], rt[3], rb[3];
    world_ray(C, u + 1, v, rr); world_ray(C, u - 1, v, rl); world_ray(C, u, v - 1, rt); world_ray(C, u, v + 1, rb);
    const float dr = depth[(size_t)v * W + u + 1], dl = depth[(size_t)v * W + u - 1];
    const float dt = depth[(size_t)(v - 1) * W + u], db = depth[(size_t)(v + 1) * W + u];
    Stencil s;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        s.a[k] = dr * rr[k] - dl * rl[k];  // the camera centre cancels in the differences
        s.b[k] = dt * rt[k] - db * rb[k];
    }
    const float m0 = s.a[1] * s.b[2] - s.a[2] * s.b[1], m1 = s.a[2] * s.b[0] - s.a[0] * s.b[2], m2 = s.a[0] * s.b[1] - s.a[1] * s.b[0];
    s.len = sqrtf(m0 * m0 + m1 * m1 + m2 * m2);
    const float dn = fmaxf(s.len, 1e-12f);  // F.normalize(eps = 1e-12)
    s.n[0] = m0 / dn; s.n[1] = m1 / dn; s.n[2] = m2 / dn;
    return s;
}

__global__ void __launch_bounds__(256) sobel_normal_kernel(int W, int H, const float* __restrict__ depth,
                                                           const float* __restrict__ alpha, const float* __restrict__ bg,
                                                           const float* __restrict__ view, float fx, float fy, float cx,
                                                           float cy, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * H) return;
    const int u = p % W, v = p / W;
    float n[3] = {0.f, 0.f, 0.f};
    if (u > 0 && u < W - 1 && v > 0 && v < H - 1) {
        const CamInv C = load_cam_inv(view, fx, fy, cx, cy);
        const Stencil s = eval_stencil(C, depth, W, u, v);
        n[0] = s.n[0]; n[1] = s.n[1]; n[2] = s.n[2];
    }
    const float al = alpha[p];
    const size_t N = (size_t)W * H;
#pragma unroll
    for (int c = 0; c < 3; c++) out[c * N + p] = n[c] * al + bg[c] * (1.f - al);
}

// gradients of the stencil at (u, v) with respect to a and b, given the upstream gradient of its normal
__device__ __forceinline__ void stencil_grads(const CamInv& C, const float* __restrict__ depth,
                                              const float* __restrict__ alpha, const float* __restrict__ g, int W, int H,
                                              int u, int v, float da[3], float db[3], float n[3]) {
    da[0] = da[1] = da[2] = db[0] = db[1] = db[2] = n[0] = n[1] = n[2] = 0.f;
    if (!(u > 0 && u < W - 1 && v > 0 && v < H - 1)) return;
    const size_t N = (size_t)W * H, q = (size_t)v * W + u;
    const Stencil s = eval_stencil(C, depth, W, u, v);
    n[0] = s.n[0]; n[1] = s.n[1]; n[2] = s.n[2];
    const float al = alpha[q];
    const float dn[3] = {g[q] * al, g[N + q] * al, g[2 * N + q] * al};
    float dm[3];
    if (s.len >= 1e-12f) {
        const float ndn = s.n[0] * dn[0] + s.n[1] * dn[1] + s.n[2] * dn[2];
#pragma unroll
        for (int k = 0; k < 3; k++) dm[k] = (dn[k] - s.n[k] * ndn) / s.len;
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) dm[k] = dn[k] / 1e-12f;
    }
    // m = a x b:  dL/da = b x dm,  dL/db = dm x a
    da[0] = s.b[1] * dm[2] - s.b[2] * dm[1]; da[1] = s.b[2] * dm[0] - s.b[0] * dm[2]; da[2] = s.b[0] * dm[1] - s.b[1] * dm[0];
    db[0] = dm[1] * s.a[2] - dm[2] * s.a[1]; db[1] = dm[2] * s.a[0] - dm[0] * s.a[2]; db[2] = dm[0] * s.a[1] - dm[1] * s.a[0];
}

// gather form: pixel p is the right neighbour of (u-1, v), the left of (u+1, v), the top of (u, v+1), the bottom of
// (u, v-1); no atomics, bitwise reproducible.  A 16x16-pixel workgroup evaluates the stencil gradients of its tile plus a
// one-pixel halo ONCE into LDS (324 stencils for 256 pixels; every stencil costs ~15 divisions and a square root, and a
// thread that evaluated its four neighbours itself paid for five) and every pixel gathers its four terms from there.
constexpr int SOB_T = 16, SOB_H = SOB_T + 2;
__global__ void __launch_bounds__(SOB_T* SOB_T) sobel_normal_bwd_kernel(int W, int H, const float* __restrict__ depth,
                                                                       const float* __restrict__ alpha,
                                                                       const float* __restrict__ bg, const float* __restrict__ view,
                                                                       float fx, float fy, float cx, float cy,
                                                                       const float* __restrict__ g, float* __restrict__ d_depth,
                                                                       float* __restrict__ d_alpha) {
    __shared__ float s_da[3][SOB_H * SOB_H], s_db[3][SOB_H * SOB_H], s_n[3][SOB_H * SOB_H];
    const int x0 = blockIdx.x * SOB_T, y0 = blockIdx.y * SOB_T;
    const CamInv C = load_cam_inv(view, fx, fy, cx, cy);
    for (int h = threadIdx.x; h < SOB_H * SOB_H; h += SOB_T * SOB_T) {
        const int hu = x0 - 1 + h % SOB_H, hv = y0 - 1 + h / SOB_H;
        float da[3] = {0.f, 0.f, 0.f}, db[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 0.f};
        if (hu >= 0 && hu < W && hv >= 0 && hv < H) stencil_grads(C, depth, alpha, g, W, H, hu, hv, da, db, n);
#pragma unroll
        for (int k = 0; k < 3; k++) { s_da[k][h] = da[k]; s_db[k][h] = db[k]; s_n[k][h] = n[k]; }
    }
    __syncthreads();
    const int lx = threadIdx.x % SOB_T, ly = threadIdx.x / SOB_T;
    const int u = x0 + lx, v = y0 + ly;
    if (u >= W || v >= H) return;
    const size_t N = (size_t)W * H, p = (size_t)v * W + u;
    const int hc = (ly + 1) * SOB_H + lx + 1;  // this pixel in the halo grid
    d_alpha[p] = g[p] * (s_n[0][hc] - bg[0]) + g[N + p] * (s_n[1][hc] - bg[1]) + g[2 * N + p] * (s_n[2][hc] - bg[2]);
    float acc[3] = {0.f, 0.f, 0.f};
    // positions outside the image hold zeros: adding them changes nothing
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (u > 0) acc[k] += s_da[k][hc - 1];
        if (u < W - 1) acc[k] -= s_da[k][hc + 1];
        if (v < H - 1) acc[k] += s_db[k][hc + SOB_H];
        if (v > 0) acc[k] -= s_db[k][hc - SOB_H];
    }
    float r[3];
    world_ray(C, u, v, r);
    d_depth[p] = r[0] * acc[0] + r[1] * acc[1] + r[2] * acc[2];
}

// ---------------------------------------------------------------- shading inputs of the material stage
// pbr_render, pbr/__init__.py:25-43, turns the planar G-buffer maps into the pixel-major arrays pbr_shading wants: the
// normal map re-normalised where it is non-zero, the albedo clamped to [0, 1], the roughness remapped to [rmin, rmax], the
// metallic map estimated as alpha * clamp(1 - roughness, 0, 1) when the model does not learn one, and four
// permute(1, 2, 0) copies -- ~20 framework kernels forward, ~10 backward.  One launch each way; only the albedo carries a
// gradient (through its clamp), as in the reference (normals, roughness and the estimate are detached there).
__global__ void __launch_bounds__(256) pbr_inputs_kernel(int N, const float* __restrict__ normal_map, const float* __restrict__ albedo_map,
                                                         const float* __restrict__ roughness_map, const float* __restrict__ alpha_map,
                                                         const float* __restrict__ metallic_map, float rmin, float rmax,
                                                         float* __restrict__ normals, float* __restrict__ albedo,
                                                         float* __restrict__ roughness, float* __restrict__ metallic) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const size_t n = (size_t)N;
    float n0 = normal_map[p], n1 = normal_map[n + p], n2 = normal_map[2 * n + p];
    const float len = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);  // torch.norm(dim=0); F.normalize divides by max(norm, 1e-12)
    if (len > 0.f) { const float dn = fmaxf(len, 1e-12f); n0 /= dn; n1 /= dn; n2 /= dn; }
    normals[3 * (size_t)p] = n0; normals[3 * (size_t)p + 1] = n1; normals[3 * (size_t)p + 2] = n2;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float a = albedo_map[c * n + p];
        albedo[3 * (size_t)p + c] = a < 0.f ? 0.f : (a > 1.f ? 1.f : a);
    }
    const float r = roughness_map[p];
    if (metallic_map != nullptr) metallic[p] = metallic_map[p];
    else {
        const float q = 1.0f - r;
        metallic[p] = alpha_map[p] * (q < 0.f ? 0.f : (q > 1.f ? 1.f : q));
    }
    roughness[p] = r * (rmax - rmin) + rmin;
}
__global__ void __launch_bounds__(256) pbr_inputs_bwd_kernel(int N, const float* __restrict__ albedo_map, const float* __restrict__ d_albedo,
                                                             float* __restrict__ d_albedo_map) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const size_t n = (size_t)N;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float a = albedo_map[c * n + p];
        d_albedo_map[c * n + p] = (a >= 0.f && a <= 1.f) ? d_albedo[3 * (size_t)p + c] : 0.f;  // clamp's backward
    }
}

// ---------------------------------------------------------------- parameter activations (GM:113-144 getters)
// scales = exp(_scaling), rotations = _rotation / max(|_rotation|, 1e-12), opacity / albedo / roughness / metallic =
// sigmoid(raw): six getters, ~9 PyTorch launches forward and ~14 backward per view; one launch each way here.
struct ActPtrs {
    const float* scaling; const float* rotation; const float* opacity; const float* albedo; const float* roughness; const float* metallic;
    float* scales; float* rotations; float* opacities; float* albedo_a; float* roughness_a; float* metallic_a;
};
struct ActGradPtrs {
    const float* rotation;  // raw quaternion
    const float* scales; const float* opacities; const float* albedo_a; const float* roughness_a; const float* metallic_a;  // activated (saved outputs)
    const float* d_scales; const float* d_rotations; const float* d_opacities; const float* d_albedo_a; const float* d_roughness_a; const float* d_metallic_a;
    float* d_scaling; float* d_rotation; float* d_opacity; float* d_albedo; float* d_roughness; float* d_metallic;
};

__device__ __forceinline__ float act_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }  // at::sigmoid: 1 / (1 + exp(-x))
__device__ __forceinline__ float act_sigmoid_bwd(float g, float y) { return g * (1.f - y) * y; }

// Both kernels request ALL of a thread's inputs before the first result is stored: written group by group (load, compute,
// store, next group) every group is a memory round trip of its own -- the stores may alias the later loads as far as
// the compiler knows -- and a one-pass kernel over 64 B per Gaussian then costs six latencies instead of one.
__global__ void __launch_bounds__(256) activate_kernel(int P, ActPtrs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float s[3] = {0.f, 0.f, 0.f}, al[3] = {0.f, 0.f, 0.f}, op = 0.f, ro = 0.f, me = 0.f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scaling != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) s[c] = a.scaling[3 * i + c];
    }
    if (a.rotation != nullptr) q = reinterpret_cast<const float4*>(a.rotation)[i];
    if (a.opacity != nullptr) op = a.opacity[i];
    if (a.albedo != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) al[c] = a.albedo[3 * i + c];
    }
    if (a.roughness != nullptr) ro = a.roughness[i];
    if (a.metallic != nullptr) me = a.metallic[i];
    if (a.scaling != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.scales[3 * i + c] = expf(s[c]);
    }
    if (a.rotation != nullptr) {  // F.normalize(p=2, dim=1, eps=1e-12)
        const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        reinterpret_cast<float4*>(a.rotations)[i] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
    }
    if (a.opacity != nullptr) a.opacities[i] = act_sigmoid(op);
    if (a.albedo != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.albedo_a[3 * i + c] = act_sigmoid(al[c]);
    }
    if (a.roughness != nullptr) a.roughness_a[i] = act_sigmoid(ro);
    if (a.metallic != nullptr) a.metallic_a[i] = act_sigmoid(me);
}

__global__ void __launch_bounds__(256) activate_bwd_kernel(int P, ActGradPtrs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float gs[3] = {0.f, 0.f, 0.f}, ys[3] = {0.f, 0.f, 0.f}, ga[3] = {0.f, 0.f, 0.f}, ya[3] = {0.f, 0.f, 0.f};
    float go = 0.f, yo = 0.f, gr = 0.f, yr = 0.f, gm = 0.f, ym = 0.f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f), g = q;
    if (a.d_scaling != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) { gs[c] = a.d_scales[3 * i + c]; ys[c] = a.scales[3 * i + c]; }
    }
    if (a.d_rotation != nullptr) {
        q = reinterpret_cast<const float4*>(a.rotation)[i];
        g = reinterpret_cast<const float4*>(a.d_rotations)[i];
    }
    if (a.d_opacity != nullptr) { go = a.d_opacities[i]; yo = a.opacities[i]; }
    if (a.d_albedo != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) { ga[c] = a.d_albedo_a[3 * i + c]; ya[c] = a.albedo_a[3 * i + c]; }
    }
    if (a.d_roughness != nullptr) { gr = a.d_roughness_a[i]; yr = a.roughness_a[i]; }
    if (a.d_metallic != nullptr) { gm = a.d_metallic_a[i]; ym = a.metallic_a[i]; }
    if (a.d_scaling != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.d_scaling[3 * i + c] = gs[c] * ys[c];
    }
    if (a.d_rotation != nullptr) {
        const float nn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        const float n = fmaxf(nn, 1e-12f);
        // y = q / n: dq = g / n - q (g . q) / (n^2 |q|); below the eps clamp the denominator is a constant
        const float dot = g.x * q.x + g.y * q.y + g.z * q.z + g.w * q.w;
        const float k = nn >= 1e-12f ? dot / (n * n * nn) : 0.f;
        reinterpret_cast<float4*>(a.d_rotation)[i] = make_float4(g.x / n - q.x * k, g.y / n - q.y * k, g.z / n - q.z * k, g.w / n - q.w * k);
    }
    if (a.d_opacity != nullptr) a.d_opacity[i] = act_sigmoid_bwd(go, yo);
    if (a.d_albedo != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.d_albedo[3 * i + c] = act_sigmoid_bwd(ga[c], ya[c]);
    }
    if (a.d_roughness != nullptr) a.d_roughness[i] = act_sigmoid_bwd(gr, yr);
    if (a.d_metallic != nullptr) a.d_metallic[i] = act_sigmoid_bwd(gm, ym);
}

}  // namespace

extern "C" {

int gs2m_pack_features_forward(int P, const float* xyz, const float* scales, const float* rotations, const float* albedo,
                               const float* roughness, const float* metallic, const float* campos,
                               const float* view, int z_depth, int blend_metallic, float* features, void* stream) {
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if (P == 0) return GS2M_OK;
    if (!xyz || !scales || !rotations || !albedo || !roughness || !metallic || !campos || !view || !features)
        return GS2M_ERR_INVALID_ARG;
    pack_features_kernel<<<(P + 255) / 256, 256, 0, (hipStream_t)stream>>>(P, xyz, scales, rotations, albedo, roughness,
                                                                          metallic, campos, view, z_depth,
                                                                          blend_metallic, features);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_pack_features_backward(int P, const float* xyz, const float* scales, const float* rotations, const float* campos,
                                const float* view, int z_depth, int blend_metallic, const float* dL_dfeatures,
                                float* dL_dxyz, float* dL_drotations, float* dL_dalbedo, float* dL_droughness,
                                float* dL_dmetallic, void* stream) {
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if (P == 0) return GS2M_OK;
    if (!xyz || !scales || !rotations || !campos || !view || !dL_dfeatures || !dL_dxyz || !dL_drotations || !dL_dalbedo ||
        !dL_droughness || !dL_dmetallic)
        return GS2M_ERR_INVALID_ARG;
    pack_features_bwd_kernel<<<(P + 255) / 256, 256, 0, (hipStream_t)stream>>>(
        P, xyz, scales, rotations, campos, view, z_depth, blend_metallic, dL_dfeatures, dL_dxyz, dL_drotations,
        dL_dalbedo, dL_droughness, dL_dmetallic);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_gbuffer_post_forward(int width, int height, const float* buffer, const float* rays, const float* view,
                              int z_depth, uint8_t* normal_mask, float* local_normal_map, float* depth_map, void* stream) {
    if (width <= 0 || height <= 0 || !buffer || !view || !normal_mask || !local_normal_map || !depth_map)
        return GS2M_ERR_INVALID_ARG;
    if (!z_depth && !rays) return GS2M_ERR_INVALID_ARG;
    const int N = width * height;
    gbuffer_post_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, buffer, rays, view, z_depth,
                                                                         normal_mask, local_normal_map, depth_map);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_gbuffer_post_backward(int width, int height, const float* buffer, const float* rays, const float* view,
                               int z_depth, const float* dL_dlocal_normal, const float* dL_ddepth, float* dL_dbuffer,
                               void* stream) {
    if (width <= 0 || height <= 0 || !buffer || !view || !dL_dbuffer) return GS2M_ERR_INVALID_ARG;
    if (!z_depth && !rays) return GS2M_ERR_INVALID_ARG;
    const int N = width * height;
    GBufDirect dir = {};
    gbuffer_post_bwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, buffer, rays, view, z_depth,
                                                                             dL_dlocal_normal, dL_ddepth, dir, dL_dbuffer);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_gbuffer_maps_backward(int width, int height, const float* buffer, const float* rays, const float* view,
                               int z_depth, const float* dL_dlocal_normal, const float* dL_ddepth, const float* dL_dalpha,
                               const float* dL_ddistance, const float* dL_dnormal, const float* dL_dalbedo,
                               const float* dL_droughness, const float* dL_dmetallic, float* dL_dbuffer, void* stream) {
    if (width <= 0 || height <= 0 || !buffer || !view || !dL_dbuffer) return GS2M_ERR_INVALID_ARG;
    if (!z_depth && !rays) return GS2M_ERR_INVALID_ARG;
    const int N = width * height;
    const size_t n = (size_t)N;
    GBufDirect dir = {};
    dir.all = 1;
    dir.g[0] = dL_dalpha;
    dir.g[1] = dL_ddistance;
    for (int c = 0; c < 3; c++) dir.g[2 + c] = dL_dnormal ? dL_dnormal + c * n : nullptr;
    for (int c = 0; c < 3; c++) dir.g[5 + c] = dL_dalbedo ? dL_dalbedo + c * n : nullptr;
    dir.g[8] = dL_droughness;
    dir.g[9] = dL_dmetallic;
    gbuffer_post_bwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, buffer, rays, view, z_depth,
                                                                             dL_dlocal_normal, dL_ddepth, dir, dL_dbuffer);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_sobel_normal_forward(int width, int height, const float* depth, const float* alpha, const float* bg,
                              const float* view, float fx, float fy, float cx, float cy, float* sobel_map, void* stream) {
    if (width <= 0 || height <= 0 || !depth || !alpha || !bg || !view || !sobel_map || fx == 0.f || fy == 0.f)
        return GS2M_ERR_INVALID_ARG;
    const int N = width * height;
    sobel_normal_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(width, height, depth, alpha, bg, view, fx, fy, cx, cy,
                                                                         sobel_map);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_sobel_normal_backward(int width, int height, const float* depth, const float* alpha, const float* bg,
                               const float* view, float fx, float fy, float cx, float cy, const float* dL_dsobel,
                               float* dL_ddepth, float* dL_dalpha, void* stream) {
    if (width <= 0 || height <= 0 || !depth || !alpha || !bg || !view || !dL_dsobel || !dL_ddepth || !dL_dalpha || fx == 0.f ||
        fy == 0.f)
        return GS2M_ERR_INVALID_ARG;
    const dim3 grid((width + SOB_T - 1) / SOB_T, (height + SOB_T - 1) / SOB_T);
    sobel_normal_bwd_kernel<<<grid, SOB_T * SOB_T, 0, (hipStream_t)stream>>>(width, height, depth, alpha, bg, view, fx, fy, cx,
                                                                            cy, dL_dsobel, dL_ddepth, dL_dalpha);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_pbr_inputs_forward(int width, int height, const float* normal_map, const float* albedo_map, const float* roughness_map,
                            const float* alpha_map, const float* metallic_map, float min_roughness, float max_roughness,
                            float* normals, float* albedo, float* roughness, float* metallic, void* stream) {
    if (width <= 0 || height <= 0 || !normal_map || !albedo_map || !roughness_map || (!metallic_map && !alpha_map) || !normals || !albedo ||
        !roughness || !metallic)
        return GS2M_ERR_INVALID_ARG;
    const int N = width * height;
    pbr_inputs_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, normal_map, albedo_map, roughness_map, alpha_map, metallic_map,
                                                                       min_roughness, max_roughness, normals, albedo, roughness, metallic);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_pbr_inputs_backward(int width, int height, const float* albedo_map, const float* dL_dalbedo, float* dL_dalbedo_map, void* stream) {
    if (width <= 0 || height <= 0 || !albedo_map || !dL_dalbedo || !dL_dalbedo_map) return GS2M_ERR_INVALID_ARG;
    const int N = width * height;
    pbr_inputs_bwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(N, albedo_map, dL_dalbedo, dL_dalbedo_map);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_activate_forward(int P, const float* scaling, const float* rotation, const float* opacity, const float* albedo,
                          const float* roughness, const float* metallic, float* scales, float* rotations, float* opacities,
                          float* albedo_a, float* roughness_a, float* metallic_a, void* stream) {
    if (P == 0) return GS2M_OK;
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if ((scaling && !scales) || (rotation && !rotations) || (opacity && !opacities) || (albedo && !albedo_a) ||
        (roughness && !roughness_a) || (metallic && !metallic_a))
        return GS2M_ERR_INVALID_ARG;
    if (rotation && ((((uintptr_t)rotation) | ((uintptr_t)rotations)) & 15)) return GS2M_ERR_UNSUPPORTED;
    ActPtrs a = {scaling, rotation, opacity, albedo, roughness, metallic, scales, rotations, opacities, albedo_a, roughness_a, metallic_a};
    activate_kernel<<<(P + 255) / 256, 256, 0, (hipStream_t)stream>>>(P, a);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_activate_backward(int P, const float* rotation, const float* scales, const float* opacities, const float* albedo_a,
                           const float* roughness_a, const float* metallic_a, const float* dL_dscales,
                           const float* dL_drotations, const float* dL_dopacities, const float* dL_dalbedo_a,
                           const float* dL_droughness_a, const float* dL_dmetallic_a, float* dL_dscaling, float* dL_drotation,
                           float* dL_dopacity, float* dL_dalbedo, float* dL_droughness, float* dL_dmetallic, void* stream) {
    if (P == 0) return GS2M_OK;
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if ((dL_dscaling && !(dL_dscales && scales)) || (dL_drotation && !(dL_drotations && rotation)) ||
        (dL_dopacity && !(dL_dopacities && opacities)) || (dL_dalbedo && !(dL_dalbedo_a && albedo_a)) ||
        (dL_droughness && !(dL_droughness_a && roughness_a)) || (dL_dmetallic && !(dL_dmetallic_a && metallic_a)))
        return GS2M_ERR_INVALID_ARG;
    if (dL_drotation && ((((uintptr_t)rotation) | ((uintptr_t)dL_drotations) | ((uintptr_t)dL_drotation)) & 15)) return GS2M_ERR_UNSUPPORTED;
    ActGradPtrs a = {rotation, scales, opacities, albedo_a, roughness_a, metallic_a, dL_dscales, dL_drotations, dL_dopacities,
                     dL_dalbedo_a, dL_droughness_a, dL_dmetallic_a, dL_dscaling, dL_drotation, dL_dopacity, dL_dalbedo,
                     dL_droughness, dL_dmetallic};
    activate_bwd_kernel<<<(P + 255) / 256, 256, 0, (hipStream_t)stream>>>(P, a);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

}  // extern "C"
