// Photometric term of the multi-view loss for gfx950: the plane-induced patch warp + NCC of
// utils/loss_utils.py:303-349 (multi_view_loss) as one kernel each way.
//
// Per sampled pixel (up to 102,400 of them, `multi_view_sample_num`) the reference builds a 3x3 homography from the
// rendered plane (camera-space normal n, distance d) and the relative pose,
//     H = K_near (R_rn - t_rn n^T / d) K_ref^-1                                     (:322-327)
// warps the (2P+1)^2 patch positions around the pixel (:309-311, _patch_warp :456-466), samples the reference and the
// neighbour grey images bilinearly (F.grid_sample, align_corners=True, zero padding, :317, :333) and scores the two
// patches with 1 - NCC^2 (_loss_ncc :468-509).  In PyTorch that is two batched 3x3 matmuls over N matrices (BLAS at
// K = 3), an einsum, two grid_samples over 49 N points and five 7x7 "ones" conv2d that are just sums, plus their
// backward -- here eight lanes per sample keep the five running sums in registers:
//     h = M p - b (n . r) / d,   M = K_near R_rn K_ref^-1,  b = K_near t_rn,  r = K_ref^-1 p      (p = (x, y, 1))
// and the backward recomputes the samples, differentiates the bilinear lookups with respect to the warped position and
// chains through h to n and d (the reference image, the pixel positions and the poses get no gradient, as there).
#include "common.h"
#include "../../include/gs2m_mvs.h"
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

namespace {

// ---------------------------------------------------------------- deterministic scatter (round 6)
// The two bilinear backwards below scatter every sample's gradient into four texels.  With fp32 atomics the order of the
// additions -- and so the last bits of every texel, and over a few hundred training iterations the Gaussian count of a run --
// differs from run to run.  Deterministic mode (the default; gs2m_mvs_set_deterministic): two passes over the samples, the first for
// the largest magnitude that will be scattered (an integer atomicMax on the float's bits: order-independent; mv_geo's first pass also
// does all the rest of its backward and parks each sample's contributions, so the chain is evaluated once), the second adding
// llrint(value x 2^shift) into 64-bit integers -- integer addition is associative, so the sum does not depend on the order --
// with shift chosen from that maximum so that as many contributions as the call has samples cannot overflow; a third kernel
// adds the sums, converted back, to the output and clears the integers for the next call.  Resolution: 2^-40 of the largest
// contribution or better (the float atomics round every addition to 2^-24 of the running sum).
enum { SCATTER_FLOAT = 0, SCATTER_MAX = 1, SCATTER_FIXED = 2 };
struct DetScale {
    const uint32_t* maxbits;  // float bits of the largest |value| (SCATTER_MAX pass)
    int headroom;             // bits kept free for the number of contributions
};
__device__ __forceinline__ int det_shift(const DetScale d) {
    const uint32_t b = *d.maxbits;
    const int e = (int)((b >> 23) & 0xFFu) - 127;  // largest |value| < 2^(e + 1)
    return 62 - d.headroom - (e + 1);
}
__device__ __forceinline__ void det_note_max(uint32_t* maxbits, float v) {  // one atomic per wave
    uint32_t m = __float_as_uint(fabsf(v));
    if (m >= 0x7F800000u) m = 0u;  // inf / NaN: nothing sensible to scale by (the float path would poison the texel as well)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(maxbits, m);
}
__device__ __forceinline__ void det_add(long long* acc, float v, double scale) {
    if (v != 0.f && fabsf(v) < __builtin_inff()) atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)__double2ll_rn((double)v * scale));
}
__global__ void __launch_bounds__(256) det_finalize_kernel(size_t n, long long* __restrict__ acc, DetScale d, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long a = acc[i];
    if (a != 0) {
        out[i] += (float)((double)a * ldexp(1.0, -det_shift(d)));
        acc[i] = 0;
    }
}

struct NccConst {
    float M[9];     // K_near R_rn K_ref^-1, row major
    float b[3];     // K_near t_rn
    float Kinv[9];  // K_ref^-1 at the NCC scale, row major
    float inv_scale;  // 1 / ncc_scale: full-resolution pixel -> grey-image pixel
    int P;          // patch half size
    int w, h;       // grey images
};

struct Bilinear {
    float v, dx, dy;  // value and its derivative with respect to the sampling position
};

// F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) at pixel position (x, y)
__device__ __forceinline__ Bilinear sample_zero(const float* __restrict__ img, int w, int h, float x, float y) {
    Bilinear r = {0.f, 0.f, 0.f};
    if (!(x > -1.f && x < (float)w && y > -1.f && y < (float)h)) return r;  // also rejects NaN: all four texels outside
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const float fx = x - xf, fy = y - yf;
    auto at = [&](int xx, int yy) { return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? img[(size_t)yy * w + xx] : 0.f; };
    const float v00 = at(x0, y0), v10 = at(x0 + 1, y0), v01 = at(x0, y0 + 1), v11 = at(x0 + 1, y0 + 1);
    r.v = v00 * (1.f - fx) * (1.f - fy) + v10 * fx * (1.f - fy) + v01 * (1.f - fx) * fy + v11 * fx * fy;
    r.dx = (v10 - v00) * (1.f - fy) + (v11 - v01) * fy;
    r.dy = (v01 - v00) * (1.f - fx) + (v11 - v10) * fx;
    return r;
}

struct Warp {
    float qx, qy;      // warped position in the neighbour image
    float hx, hy, hz;  // homogeneous coordinates (hz includes the +1e-10 of _patch_warp)
    float rx, ry, rz;  // K_ref^-1 p
    float s;           // n . r
};

__device__ __forceinline__ Warp warp_point(const NccConst& C, float px, float py, const float (&n)[3], float inv_d) {
    Warp W;
    W.rx = C.Kinv[0] * px + C.Kinv[1] * py + C.Kinv[2];
    W.ry = C.Kinv[3] * px + C.Kinv[4] * py + C.Kinv[5];
    W.rz = C.Kinv[6] * px + C.Kinv[7] * py + C.Kinv[8];
    W.s = n[0] * W.rx + n[1] * W.ry + n[2] * W.rz;
    const float k = W.s * inv_d;
    W.hx = C.M[0] * px + C.M[1] * py + C.M[2] - C.b[0] * k;
    W.hy = C.M[3] * px + C.M[4] * py + C.M[5] - C.b[1] * k;
    W.hz = C.M[6] * px + C.M[7] * py + C.M[8] - C.b[2] * k + 1e-10f;
    W.qx = W.hx / W.hz;
    W.qy = W.hy / W.hz;
    return W;
}

struct Sums {
    float r, n, rr, nn, rn;
};

// Eight lanes share one sample, one patch row each (a sample alone in a thread is a chain of ~100 dependent gathers, and
// 102,400 samples are only 1.5 waves per SIMD: the kernel ran at the latency of that chain); the five moments -- and in the
// backward the four gradient components -- are summed across the eight lanes.
constexpr int NCC_LPS = 8;

__device__ __forceinline__ float group_sum8(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    return v;
}

template <bool BWD>
__global__ void __launch_bounds__(128) patch_ncc_kernel(int N, NccConst C, const float* __restrict__ pixels,
                                                        const float* __restrict__ normals, const float* __restrict__ dists,
                                                        const float* __restrict__ ref_gray, const float* __restrict__ near_gray,
                                                        float* __restrict__ ncc_out, const float* __restrict__ d_ncc,
                                                        float* __restrict__ d_normals, float* __restrict__ d_dists) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i_raw = gid / NCC_LPS, sub = gid % NCC_LPS;
    const bool live = i_raw < N;
    const int i = live ? i_raw : N - 1;  // whole groups stay converged for the shuffles
    const float cx = pixels[2 * (size_t)i] * C.inv_scale, cy = pixels[2 * (size_t)i + 1] * C.inv_scale;
    const float n[3] = {normals[3 * (size_t)i], normals[3 * (size_t)i + 1], normals[3 * (size_t)i + 2]};
    const float d = dists[i], inv_d = 1.0f / d;
    const float tps = (float)((2 * C.P + 1) * (2 * C.P + 1));
    // The variances and the covariance are differences of nearly equal sums (grey values ~0.5, contrast ~0.05): summing
    // the raw moments in fp32 costs three digits.  They are shift invariant, so both patches are accumulated relative to
    // their centre samples.
    const float r0 = sample_zero(ref_gray, C.w, C.h, cx, cy).v;
    float v0;
    {
        const Warp W = warp_point(C, cx, cy, n, inv_d);
        v0 = sample_zero(near_gray, C.w, C.h, W.qx, W.qy).v;
    }
    Sums S = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int oy = -C.P + sub; oy <= C.P; oy += NCC_LPS)
        for (int ox = -C.P; ox <= C.P; ox++) {
            const float px = cx + (float)ox, py = cy + (float)oy;
            const float r = sample_zero(ref_gray, C.w, C.h, px, py).v - r0;
            const Warp W = warp_point(C, px, py, n, inv_d);
            const float v = sample_zero(near_gray, C.w, C.h, W.qx, W.qy).v - v0;
            S.r += r; S.n += v; S.rr += r * r; S.nn += v * v; S.rn += r * v;
        }
    S.r = group_sum8(S.r); S.n = group_sum8(S.n); S.rr = group_sum8(S.rr); S.nn = group_sum8(S.nn); S.rn = group_sum8(S.rn);
    const float ref_avg = S.r / tps, nea_avg = S.n / tps;
    const float cross = S.rn - nea_avg * S.r;
    const float ref_var = S.rr - ref_avg * S.r;
    const float nea_var = S.nn - nea_avg * S.n;
    const float D = ref_var * nea_var + 1e-8f;
    const float raw = 1.0f - cross * cross / D;
    if (!BWD) {
        if (live && sub == 0) ncc_out[i] = fminf(fmaxf(raw, 0.0f), 2.0f);
        return;
    }
    // d ncc / d v_k for the neighbour samples v_k; the clamp passes the gradient on [0, 2]
    const float g = (raw >= 0.0f && raw <= 2.0f) ? -d_ncc[i] : 0.0f;                // d / d cc
    const float g_cross = g * 2.0f * cross / D, g_var = -g * cross * cross * ref_var / (D * D);
    float dn[3] = {0.f, 0.f, 0.f}, dd = 0.f;
    if (g != 0.0f) {
        for (int oy = -C.P + sub; oy <= C.P; oy += NCC_LPS)
            for (int ox = -C.P; ox <= C.P; ox++) {
                const float px = cx + (float)ox, py = cy + (float)oy;
                const float r = sample_zero(ref_gray, C.w, C.h, px, py).v - r0;
                const Warp W = warp_point(C, px, py, n, inv_d);
                const Bilinear B = sample_zero(near_gray, C.w, C.h, W.qx, W.qy);
                const float gv = g_cross * (r - ref_avg) + g_var * (2.0f * (B.v - v0) - 2.0f * nea_avg);
                const float gqx = gv * B.dx, gqy = gv * B.dy;
                // q = (hx, hy) / hz
                const float ghx = gqx / W.hz, ghy = gqy / W.hz, ghz = -(gqx * W.qx + gqy * W.qy) / W.hz;
                const float gb = ghx * C.b[0] + ghy * C.b[1] + ghz * C.b[2];       // h = M p - b (n . r) / d
                const float k = -gb * inv_d;
                dn[0] += k * W.rx; dn[1] += k * W.ry; dn[2] += k * W.rz;
                dd += gb * W.s * inv_d * inv_d;
            }
    }
    dn[0] = group_sum8(dn[0]); dn[1] = group_sum8(dn[1]); dn[2] = group_sum8(dn[2]); dd = group_sum8(dd);
    if (live && sub == 0) {
        d_normals[3 * (size_t)i] = dn[0]; d_normals[3 * (size_t)i + 1] = dn[1]; d_normals[3 * (size_t)i + 2] = dn[2];
        d_dists[i] = dd;
    }
}

// ---------------------------------------------------------------- roughness_loss variant (utils/loss_utils.py:138-243)
// Forward only (the reference evaluates it under no_grad): besides the grey-value NCC it needs the NCC of the Sobel
// gradient magnitudes of the two patches (_patch_gradient :232-238, 3x3 Sobel with zero padding on the patch) and the
// reference patch's variance for the low-texture switch (:209-211).  The patches are staged in LDS (49 values x 2 per
// thread), the rest is as above.
constexpr int ROUGH_THREADS = 64;
constexpr int ROUGH_MAX = 49;

__device__ __forceinline__ float ncc_from_sums(const Sums& S, float tps, float* ref_var_out) {
    const float ref_avg = S.r / tps, nea_avg = S.n / tps;
    const float cross = S.rn - nea_avg * S.r;
    const float ref_var = S.rr - ref_avg * S.r;
    const float nea_var = S.nn - nea_avg * S.n;
    if (ref_var_out != nullptr) *ref_var_out = ref_var;
    return fminf(fmaxf(1.0f - cross * cross / (ref_var * nea_var + 1e-8f), 0.0f), 2.0f);
}

__global__ void __launch_bounds__(ROUGH_THREADS) patch_ncc_rough_kernel(int N, NccConst C, const float* __restrict__ pixels,
                                                                        const float* __restrict__ normals, const float* __restrict__ dists,
                                                                        const float* __restrict__ ref_gray, const float* __restrict__ near_gray,
                                                                        float* __restrict__ ncc_gray, float* __restrict__ ncc_grad,
                                                                        float* __restrict__ ref_var) {
    __shared__ float s_r[ROUGH_MAX][ROUGH_THREADS], s_v[ROUGH_MAX][ROUGH_THREADS];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, t = threadIdx.x;
    if (i >= N) return;
    const float cx = pixels[2 * (size_t)i] * C.inv_scale, cy = pixels[2 * (size_t)i + 1] * C.inv_scale;
    const float n[3] = {normals[3 * (size_t)i], normals[3 * (size_t)i + 1], normals[3 * (size_t)i + 2]};
    const float inv_d = 1.0f / dists[i];
    const int ps = 2 * C.P + 1;
    const float tps = (float)(ps * ps);
    const float r0 = sample_zero(ref_gray, C.w, C.h, cx, cy).v;
    float v0;
    {
        const Warp W = warp_point(C, cx, cy, n, inv_d);
        v0 = sample_zero(near_gray, C.w, C.h, W.qx, W.qy).v;
    }
    Sums S = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ps; a++)
        for (int b = 0; b < ps; b++) {
            const float px = cx + (float)(a - C.P), py = cy + (float)(b - C.P);
            const float rr = sample_zero(ref_gray, C.w, C.h, px, py).v;
            const Warp W = warp_point(C, px, py, n, inv_d);
            const float vv = sample_zero(near_gray, C.w, C.h, W.qx, W.qy).v;
            s_r[a * ps + b][t] = rr; s_v[a * ps + b][t] = vv;
            const float r = rr - r0, v = vv - v0;
            S.r += r; S.n += v; S.rr += r * r; S.nn += v * v; S.rn += r * v;
        }
    float rv;
    ncc_gray[i] = ncc_from_sums(S, tps, &rv);
    ref_var[i] = rv;
    // Sobel magnitude of each patch (zero padding at the patch border), then the same statistic on the magnitudes
    auto at = [&](const float (*P)[ROUGH_THREADS], int a, int b) { return (a >= 0 && a < ps && b >= 0 && b < ps) ? P[a * ps + b][t] : 0.f; };
    auto sobel = [&](const float (*P)[ROUGH_THREADS], int a, int b) {
        const float gx = (at(P, a - 1, b + 1) - at(P, a - 1, b - 1)) + 2.f * (at(P, a, b + 1) - at(P, a, b - 1)) + (at(P, a + 1, b + 1) - at(P, a + 1, b - 1));
        const float gy = (at(P, a + 1, b - 1) - at(P, a - 1, b - 1)) + 2.f * (at(P, a + 1, b) - at(P, a - 1, b)) + (at(P, a + 1, b + 1) - at(P, a - 1, b + 1));
        return sqrtf(gx * gx + gy * gy + 1e-6f);
    };
    const float g0r = sobel(s_r, C.P, C.P), g0v = sobel(s_v, C.P, C.P);
    Sums G = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ps; a++)
        for (int b = 0; b < ps; b++) {
            const float r = sobel(s_r, a, b) - g0r, v = sobel(s_v, a, b) - g0v;
            G.r += r; G.n += v; G.rr += r * r; G.nn += v * v; G.rn += r * v;
        }
    ncc_grad[i] = ncc_from_sums(G, tps, nullptr);
}


// ---------------------------------------------------------------- bilinear lookup of the neighbour's depth / normal maps
// F.grid_sample(mode='bilinear', padding_mode='border', align_corners=True) of a (C, H, W) image at N normalised positions
// (_sample_depth_normal, utils/loss_utils.py:368-409), forward and backward (to the image AND to the positions: the neighbour
// view is rendered with gradients, and the positions depend on the rendered depth).  PyTorch's backward kernel for this op
// takes 69 ms for the 2M pixels of a 1080p view on this stack -- 90 % of multi_view_loss; with hardware fp32 atomics it is
// a ~0.1 ms scatter (one position touches four texels, and the positions are a warped pixel grid: little contention).
template <int C, bool BWD, int MODE = SCATTER_FLOAT>
__global__ void __launch_bounds__(256) grid_border_kernel(int N, int H, int W, const float* __restrict__ img, const float* __restrict__ grid,
                                                          float* __restrict__ out, const float* __restrict__ d_out,
                                                          float* __restrict__ d_img, float* __restrict__ d_grid,
                                                          long long* __restrict__ acc = nullptr, uint32_t* __restrict__ maxbits = nullptr, int headroom = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == SCATTER_MAX) {  // the largest |d_out| bounds every contribution (the bilinear weights are <= 1)
        float m = 0.f;
        if (i < N) {
#pragma unroll
            for (int c = 0; c < C; c++) m = fmaxf(m, fabsf(d_out[(size_t)i * C + c]));
        }
        det_note_max(maxbits, m);
        return;
    }
    if (i >= N) return;
    double scale = 0.0;
    if (MODE == SCATTER_FIXED) scale = ldexp(1.0, det_shift(DetScale{maxbits, headroom}));
    // unnormalise (align_corners), then clip to the border; the clip zeroes the position gradient where it binds
    float x = (grid[2 * (size_t)i] + 1.f) * 0.5f * (float)(W - 1), y = (grid[2 * (size_t)i + 1] + 1.f) * 0.5f * (float)(H - 1);
    float mx = 0.5f * (float)(W - 1), my = 0.5f * (float)(H - 1);
    if (!(x > 0.f)) { x = 0.f; mx = 0.f; } else if (x >= (float)(W - 1)) { x = (float)(W - 1); mx = 0.f; }
    if (!(y > 0.f)) { y = 0.f; my = 0.f; } else if (y >= (float)(H - 1)) { y = (float)(H - 1); my = 0.f; }
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    const float fx = x - xf, fy = y - yf;
    const bool bx = x1 <= W - 1, by = y1 <= H - 1;  // the second column / row exists (else its weight is 0 anyway)
    const float w00 = (1.f - fx) * (1.f - fy), w10 = fx * (1.f - fy), w01 = (1.f - fx) * fy, w11 = fx * fy;
    float gx = 0.f, gy = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const size_t p = (size_t)c * H * W;
        const float v00 = img[p + (size_t)y0 * W + x0];
        const float v10 = bx ? img[p + (size_t)y0 * W + x1] : 0.f;
        const float v01 = by ? img[p + (size_t)y1 * W + x0] : 0.f;
        const float v11 = (bx && by) ? img[p + (size_t)y1 * W + x1] : 0.f;
        if (!BWD) {
            out[(size_t)i * C + c] = v00 * w00 + v10 * w10 + v01 * w01 + v11 * w11;
        } else {
            const float g = d_out[(size_t)i * C + c];
            if (d_img != nullptr && g != 0.f) {
                if (MODE == SCATTER_FIXED) {
                    det_add(&acc[p + (size_t)y0 * W + x0], g * w00, scale);
                    if (bx) det_add(&acc[p + (size_t)y0 * W + x1], g * w10, scale);
                    if (by) det_add(&acc[p + (size_t)y1 * W + x0], g * w01, scale);
                    if (bx && by) det_add(&acc[p + (size_t)y1 * W + x1], g * w11, scale);
                } else {
                    unsafeAtomicAdd(&d_img[p + (size_t)y0 * W + x0], g * w00);
                    if (bx) unsafeAtomicAdd(&d_img[p + (size_t)y0 * W + x1], g * w10);
                    if (by) unsafeAtomicAdd(&d_img[p + (size_t)y1 * W + x0], g * w01);
                    if (bx && by) unsafeAtomicAdd(&d_img[p + (size_t)y1 * W + x1], g * w11);
                }
            }
            gx += g * ((v10 - v00) * (1.f - fy) + (v11 - v01) * fy);
            gy += g * ((v01 - v00) * (1.f - fx) + (v11 - v10) * fx);
        }
    }
    if (BWD && d_grid != nullptr) { d_grid[2 * (size_t)i] = gx * mx; d_grid[2 * (size_t)i + 1] = gy * my; }
}


// ---------------------------------------------------------------- geometric consistency of multi_view_loss, per pixel
// utils/loss_utils.py:256-291: back-project the pixel with the rendered depth, move it into the neighbour's camera, project,
// look the neighbour's depth / normal up there (bilinear, border), reproject with THAT depth into the reference image and
// measure the distance to the pixel (pixel_noise); compare the two normals (angle).  ~60 PyTorch launches over 2M pixels
// forward + backward; one kernel each way here.  The backward recomputes the chain and differentiates it by hand down to
// the four maps (reference depth / normal: one write per pixel; neighbour depth / normal: bilinear scatter, fp32 atomics).
struct GeoConst {
    float A[9], b[3];    // reference camera space -> neighbour camera space: Y = A P + b
    float A2[9], b2[3];  // neighbour camera space -> reference camera space
    float fx, fy, cx, cy;      // reference intrinsics
    float fxn, fyn, cxn, cyn;  // neighbour intrinsics
    int W, H, Wn, Hn;
    float occlusion;
};

struct GeoPix {
    float rx, ry, d;            // ray, depth
    float Y[3], iz, qx, qy;     // neighbour-space point, 1 / z, projection
    float zs, nraw[3], nn;      // sampled depth, sampled normal and its length
    float gzx, gzy, gnx[3], gny[3];  // d(sample)/d(q) with the border clip applied
    int x0, y0; float fx, fy; bool bx, by;  // bilinear footprint
    float s, Yp[3], Z[3], izz, ex, ey, noise;
    float nr[3], nrl, c;
    bool valid;
};

__device__ __forceinline__ GeoPix geo_eval(const GeoConst& C, int u, int v, const float* __restrict__ depth, const float* __restrict__ normal,
                                           const float* __restrict__ depth_n, const float* __restrict__ normal_n) {
    GeoPix g;
    const size_t p = (size_t)v * C.W + u, HW = (size_t)C.H * C.W, HWn = (size_t)C.Hn * C.Wn;
    g.rx = ((float)u - C.cx) / C.fx; g.ry = ((float)v - C.cy) / C.fy; g.d = depth[p];
    const float P[3] = {g.rx * g.d, g.ry * g.d, g.d};
#pragma unroll
    for (int i = 0; i < 3; i++) g.Y[i] = C.A[3 * i] * P[0] + C.A[3 * i + 1] * P[1] + C.A[3 * i + 2] * P[2] + C.b[i];
    g.iz = 1.0f / g.Y[2];
    g.qx = g.Y[0] * C.fxn * g.iz + C.cxn; g.qy = g.Y[1] * C.fyn * g.iz + C.cyn;
    g.valid = g.qx > 0.f && g.qx < (float)C.Wn && g.qy > 0.f && g.qy < (float)C.Hn && g.Y[2] > 0.1f;
    // grid_sample(border, align_corners): clip to [0, W-1]; the clip kills the position gradient where it binds
    float x = g.qx, y = g.qy, mx = 1.f, my = 1.f;
    if (!(x > 0.f)) { x = 0.f; mx = 0.f; } else if (x >= (float)(C.Wn - 1)) { x = (float)(C.Wn - 1); mx = 0.f; }
    if (!(y > 0.f)) { y = 0.f; my = 0.f; } else if (y >= (float)(C.Hn - 1)) { y = (float)(C.Hn - 1); my = 0.f; }
    const float xf = floorf(x), yf = floorf(y);
    g.x0 = (int)xf; g.y0 = (int)yf; g.fx = x - xf; g.fy = y - yf;
    g.bx = g.x0 + 1 <= C.Wn - 1; g.by = g.y0 + 1 <= C.Hn - 1;
    auto tap = [&](const float* img, float& val, float& dx, float& dy) {
        const float v00 = img[(size_t)g.y0 * C.Wn + g.x0];
        const float v10 = g.bx ? img[(size_t)g.y0 * C.Wn + g.x0 + 1] : 0.f;
        const float v01 = g.by ? img[(size_t)(g.y0 + 1) * C.Wn + g.x0] : 0.f;
        const float v11 = (g.bx && g.by) ? img[(size_t)(g.y0 + 1) * C.Wn + g.x0 + 1] : 0.f;
        val = v00 * (1.f - g.fx) * (1.f - g.fy) + v10 * g.fx * (1.f - g.fy) + v01 * (1.f - g.fx) * g.fy + v11 * g.fx * g.fy;
        dx = ((v10 - v00) * (1.f - g.fy) + (v11 - v01) * g.fy) * mx;
        dy = ((v01 - v00) * (1.f - g.fx) + (v11 - v10) * g.fx) * my;
    };
    tap(depth_n, g.zs, g.gzx, g.gzy);
#pragma unroll
    for (int c = 0; c < 3; c++) tap(normal_n + c * HWn, g.nraw[c], g.gnx[c], g.gny[c]);
    g.nn = sqrtf(g.nraw[0] * g.nraw[0] + g.nraw[1] * g.nraw[1] + g.nraw[2] * g.nraw[2]);
    g.valid = g.valid && (g.Y[2] - g.zs <= C.occlusion);
    g.s = g.zs * g.iz;
#pragma unroll
    for (int i = 0; i < 3; i++) g.Yp[i] = g.Y[i] * g.s;
#pragma unroll
    for (int i = 0; i < 3; i++) g.Z[i] = C.A2[3 * i] * g.Yp[0] + C.A2[3 * i + 1] * g.Yp[1] + C.A2[3 * i + 2] * g.Yp[2] + C.b2[i];
    g.izz = 1.0f / g.Z[2];
    g.ex = g.Z[0] * C.fx * g.izz + C.cx - (float)u;
    g.ey = g.Z[1] * C.fy * g.izz + C.cy - (float)v;
    g.noise = sqrtf(g.ex * g.ex + g.ey * g.ey);
#pragma unroll
    for (int c = 0; c < 3; c++) g.nr[c] = normal[c * HW + p];
    g.nrl = sqrtf(g.nr[0] * g.nr[0] + g.nr[1] * g.nr[1] + g.nr[2] * g.nr[2]);
    const float ir = 1.0f / (g.nrl + 1e-8f), is = 1.0f / (g.nn + 1e-8f);
    g.c = (g.nr[0] * g.nraw[0] + g.nr[1] * g.nraw[1] + g.nr[2] * g.nraw[2]) * ir * is;
    return g;
}

template <bool BWD, int MODE = SCATTER_FLOAT>
__global__ void __launch_bounds__(256) mv_geo_kernel(GeoConst C, const float* __restrict__ depth, const float* __restrict__ normal,
                                                     const float* __restrict__ depth_n, const float* __restrict__ normal_n,
                                                     float* __restrict__ noise, float* __restrict__ angle, uint8_t* __restrict__ valid,
                                                     const float* __restrict__ d_noise, const float* __restrict__ d_angle,
                                                     float* __restrict__ d_depth, float* __restrict__ d_normal,
                                                     float* __restrict__ d_depth_n, float* __restrict__ d_normal_n,
                                                     long long* __restrict__ acc = nullptr, uint32_t* __restrict__ maxbits = nullptr, int headroom = 0,
                                                     float* __restrict__ srec = nullptr /* SCATTER_MAX: 7 x n words, what each sample will scatter */) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE != SCATTER_MAX && i0 >= C.W * C.H) return;
    const int i = min(i0, C.W * C.H - 1);  // (SCATTER_MAX: whole waves stay for the wave-wide maximum; the surplus lanes note 0)
    const int u = i % C.W, v = i / C.W;
    const GeoPix g = geo_eval(C, u, v, depth, normal, depth_n, normal_n);
    const float lo = -1.0f + 1e-6f, hi = 1.0f - 1e-6f;
    const float cc = fminf(fmaxf(g.c, lo), hi);
    if (!BWD) {
        noise[i] = g.noise; angle[i] = acosf(cc); valid[i] = g.valid ? 1 : 0;
        return;
    }
    const size_t HW = (size_t)C.H * C.W, HWn = (size_t)C.Hn * C.Wn;
    const float gn = d_noise[i], ga = d_angle[i];
    float dY[3] = {0.f, 0.f, 0.f}, dzs = 0.f, dnraw[3] = {0.f, 0.f, 0.f}, dnr[3] = {0.f, 0.f, 0.f};
    if (gn != 0.f && g.noise > 0.f) {   // noise = |e|, e = project(A2 Y' + b2) - pixel
        const float dex = gn * g.ex / g.noise, dey = gn * g.ey / g.noise;
        const float dZ[3] = {dex * C.fx * g.izz, dey * C.fy * g.izz, -(dex * C.fx * g.Z[0] + dey * C.fy * g.Z[1]) * g.izz * g.izz};
        float dYp[3];
#pragma unroll
        for (int k = 0; k < 3; k++) dYp[k] = C.A2[k] * dZ[0] + C.A2[3 + k] * dZ[1] + C.A2[6 + k] * dZ[2];
        const float ds = dYp[0] * g.Y[0] + dYp[1] * g.Y[1] + dYp[2] * g.Y[2];   // Y' = Y s, s = zs / Y.z
#pragma unroll
        for (int k = 0; k < 3; k++) dY[k] += dYp[k] * g.s;
        dzs += ds * g.iz;
        dY[2] -= ds * g.zs * g.iz * g.iz;
    }
    if (ga != 0.f && g.c >= lo && g.c <= hi) {  // angle = acos(clamp(m . n_s))
        const float dc = -ga / sqrtf(1.0f - cc * cc);
        const float er = g.nrl + 1e-8f, es = g.nn + 1e-8f;
        float m[3], ns[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { m[k] = g.nr[k] / er; ns[k] = g.nraw[k] / es; }
        // y = x / (|x| + eps): dx = dy / (|x| + eps) - x (dy . x) / (|x| (|x| + eps)^2)
        const float dot_s = dc * (m[0] * g.nraw[0] + m[1] * g.nraw[1] + m[2] * g.nraw[2]);
        const float dot_r = dc * (ns[0] * g.nr[0] + ns[1] * g.nr[1] + ns[2] * g.nr[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            dnraw[k] += dc * m[k] / es - (g.nn > 0.f ? g.nraw[k] * dot_s / (g.nn * es * es) : 0.f);
            dnr[k] += dc * ns[k] / er - (g.nrl > 0.f ? g.nr[k] * dot_r / (g.nrl * er * er) : 0.f);
        }
    }
    // the lookups: scatter to the neighbour's maps, and the position gradient
    float dqx = dzs * g.gzx, dqy = dzs * g.gzy;
#pragma unroll
    for (int k = 0; k < 3; k++) { dqx += dnraw[k] * g.gnx[k]; dqy += dnraw[k] * g.gny[k]; }
    const float w00 = (1.f - g.fx) * (1.f - g.fy), w10 = g.fx * (1.f - g.fy), w01 = (1.f - g.fx) * g.fy, w11 = g.fx * g.fy;
    if (MODE == SCATTER_MAX) {
        // Deterministic mode, first of two kernels: everything but the scatter.  What this sample will add to the neighbour's maps is
        // parked (values, bilinear fractions, footprint) for mv_geo_scatter_kernel, and bounded by these four values (the weights are
        // <= 1) for the scale of the integer sums.
        const float m = fmaxf(fmaxf(fabsf(dzs), fabsf(dnraw[0])), fmaxf(fabsf(dnraw[1]), fabsf(dnraw[2])));
        det_note_max(maxbits, i0 < C.W * C.H ? m : 0.f);
        if (i0 >= C.W * C.H) return;
        const size_t n = (size_t)C.W * C.H;
        srec[i] = dzs; srec[n + i] = dnraw[0]; srec[2 * n + i] = dnraw[1]; srec[3 * n + i] = dnraw[2];
        srec[4 * n + i] = g.fx; srec[5 * n + i] = g.fy;
        reinterpret_cast<uint32_t*>(srec)[6 * n + i] = (uint32_t)g.x0 | ((uint32_t)g.y0 << 15) | (g.bx ? 1u << 30 : 0u) | (g.by ? 1u << 31 : 0u);
    } else {
        auto scatter = [&](float* img, float gv) {
            if (gv == 0.f) return;
            const size_t o = (size_t)g.y0 * C.Wn + g.x0;
            unsafeAtomicAdd(&img[o], gv * w00);
            if (g.bx) unsafeAtomicAdd(&img[o + 1], gv * w10);
            if (g.by) unsafeAtomicAdd(&img[o + C.Wn], gv * w01);
            if (g.bx && g.by) unsafeAtomicAdd(&img[o + C.Wn + 1], gv * w11);
        };
        scatter(d_depth_n, dzs);
#pragma unroll
        for (int k = 0; k < 3; k++) scatter(d_normal_n + k * HWn, dnraw[k]);
    }
    // q = (Y.x fxn / Y.z + cxn, Y.y fyn / Y.z + cyn)
    dY[0] += dqx * C.fxn * g.iz;
    dY[1] += dqy * C.fyn * g.iz;
    dY[2] -= (dqx * C.fxn * g.Y[0] + dqy * C.fyn * g.Y[1]) * g.iz * g.iz;
    float dP[3];
#pragma unroll
    for (int k = 0; k < 3; k++) dP[k] = C.A[k] * dY[0] + C.A[3 + k] * dY[1] + C.A[6 + k] * dY[2];
    d_depth[i] = dP[0] * g.rx + dP[1] * g.ry + dP[2];
#pragma unroll
    for (int k = 0; k < 3; k++) d_normal[k * HW + i] = dnr[k];
}

// Deterministic mode, second kernel: the parked contributions into the 64-bit sums (depth plane, then the three normal planes).
__global__ void __launch_bounds__(256) mv_geo_scatter_kernel(size_t n, int Wn, size_t HWn, const float* __restrict__ srec, long long* __restrict__ acc,
                                                             DetScale d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v[4] = {srec[i], srec[n + i], srec[2 * n + i], srec[3 * n + i]};
    if (v[0] == 0.f && v[1] == 0.f && v[2] == 0.f && v[3] == 0.f) return;
    const float fx = srec[4 * n + i], fy = srec[5 * n + i];
    const uint32_t pk = reinterpret_cast<const uint32_t*>(srec)[6 * n + i];
    const bool bx = (pk >> 30) & 1u, by = (pk >> 31) & 1u;
    const size_t o = (size_t)((pk >> 15) & 0x7FFFu) * Wn + (pk & 0x7FFFu);
    const float w00 = (1.f - fx) * (1.f - fy), w10 = fx * (1.f - fy), w01 = (1.f - fx) * fy, w11 = fx * fy;
    const double scale = ldexp(1.0, det_shift(d));
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long long* a = acc + (size_t)k * HWn + o;
        det_add(a, v[k] * w00, scale);
        if (bx) det_add(a + 1, v[k] * w10, scale);
        if (by) det_add(a + Wn, v[k] * w01, scale);
        if (bx && by) det_add(a + Wn + 1, v[k] * w11, scale);
    }
}

int fill(NccConst& C, const float* M, const float* b, const float* Kinv, float ncc_scale, int patch, int w, int h) {
    if (!M || !b || !Kinv || !(ncc_scale > 0.f) || patch < 0 || patch > 8 || w < 1 || h < 1) return GS2M_ERR_INVALID_ARG;
    for (int k = 0; k < 9; k++) { C.M[k] = M[k]; C.Kinv[k] = Kinv[k]; }
    for (int k = 0; k < 3; k++) C.b[k] = b[k];
    C.inv_scale = 1.0f / ncc_scale; C.P = patch; C.w = w; C.h = h;
    return GS2M_OK;
}

// ---- deterministic mode: the integer sums of one call, per (device, stream); kept all-zero between calls (det_finalize_kernel clears what it reads)
std::atomic<int> g_mvs_deterministic{1};
struct DetWorkspace {
    long long* acc = nullptr;
    size_t words = 0;
    uint32_t* maxbits = nullptr;
    float* scratch = nullptr;  // plain scratch (mv_geo's parked contributions)
    size_t scratch_words = 0;
};
std::mutex g_det_mutex;
std::map<std::pair<int, hipStream_t>, DetWorkspace> g_det_ws;
// -> the workspace for `words` 64-bit sums on this stream (grown and zeroed when needed), or nullptr when the allocation fails
DetWorkspace* det_workspace(size_t words, hipStream_t s, size_t scratch_words = 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_det_mutex);
    DetWorkspace& w = g_det_ws[std::make_pair(dev, s)];
    if (w.words < words) {
        if (w.acc != nullptr) {
            (void)hipStreamSynchronize(s);  // the smaller buffer may still be read by a finalize kernel in flight
            (void)hipFree(w.acc);
            w.acc = nullptr; w.words = 0;
        }
        if (hipMalloc((void**)&w.acc, words * sizeof(long long)) != hipSuccess) { w.acc = nullptr; return nullptr; }
        if (hipMemsetAsync(w.acc, 0, words * sizeof(long long), s) != hipSuccess) return nullptr;
        w.words = words;
    }
    if (w.maxbits == nullptr && hipMalloc((void**)&w.maxbits, 256) != hipSuccess) { w.maxbits = nullptr; return nullptr; }
    if (w.scratch_words < scratch_words) {
        if (w.scratch != nullptr) {
            (void)hipStreamSynchronize(s);
            (void)hipFree(w.scratch);
            w.scratch = nullptr; w.scratch_words = 0;
        }
        if (hipMalloc((void**)&w.scratch, scratch_words * sizeof(float)) != hipSuccess) { w.scratch = nullptr; return nullptr; }
        w.scratch_words = scratch_words;
    }
    return &w;
}
int det_headroom(long long samples) {  // bits kept free: every sample may add 4 contributions to one texel
    int b = 3;
    while ((1ll << (b - 2)) < samples && b < 40) b++;
    return b;
}

}  // namespace

extern "C" {

int gs2m_patch_ncc_forward(int N, const float* pixels, const float* normals, const float* dists, const float* ref_gray,
                           const float* near_gray, int width, int height, const float* M, const float* b, const float* Kinv,
                           float ncc_scale, int patch, float* ncc, void* stream) {
    if (N == 0) return GS2M_OK;
    if (N < 0 || !pixels || !normals || !dists || !ref_gray || !near_gray || !ncc) return GS2M_ERR_INVALID_ARG;
    NccConst C;
    const int rc = fill(C, M, b, Kinv, ncc_scale, patch, width, height);
    if (rc != GS2M_OK) return rc;
    patch_ncc_kernel<false><<<(int)(((long long)N * NCC_LPS + 127) / 128), 128, 0, (hipStream_t)stream>>>(N, C, pixels, normals, dists, ref_gray, near_gray, ncc,
                                                                             nullptr, nullptr, nullptr);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_patch_ncc_backward(int N, const float* pixels, const float* normals, const float* dists, const float* ref_gray,
                            const float* near_gray, int width, int height, const float* M, const float* b, const float* Kinv,
                            float ncc_scale, int patch, const float* dL_dncc, float* dL_dnormals, float* dL_ddists, void* stream) {
    if (N == 0) return GS2M_OK;
    if (N < 0 || !pixels || !normals || !dists || !ref_gray || !near_gray || !dL_dncc || !dL_dnormals || !dL_ddists)
        return GS2M_ERR_INVALID_ARG;
    NccConst C;
    const int rc = fill(C, M, b, Kinv, ncc_scale, patch, width, height);
    if (rc != GS2M_OK) return rc;
    patch_ncc_kernel<true><<<(int)(((long long)N * NCC_LPS + 127) / 128), 128, 0, (hipStream_t)stream>>>(N, C, pixels, normals, dists, ref_gray, near_gray, nullptr,
                                                                            dL_dncc, dL_dnormals, dL_ddists);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_patch_ncc_roughness(int N, const float* pixels, const float* normals, const float* dists, const float* ref_gray,
                             const float* near_gray, int width, int height, const float* M, const float* b, const float* Kinv,
                             float ncc_scale, int patch, float* ncc_gray, float* ncc_grad, float* ref_var, void* stream) {
    if (N == 0) return GS2M_OK;
    if (N < 0 || !pixels || !normals || !dists || !ref_gray || !near_gray || !ncc_gray || !ncc_grad || !ref_var) return GS2M_ERR_INVALID_ARG;
    if (patch > 3) return GS2M_ERR_UNSUPPORTED;
    NccConst C;
    const int rc = fill(C, M, b, Kinv, ncc_scale, patch, width, height);
    if (rc != GS2M_OK) return rc;
    patch_ncc_rough_kernel<<<(N + ROUGH_THREADS - 1) / ROUGH_THREADS, ROUGH_THREADS, 0, (hipStream_t)stream>>>(
        N, C, pixels, normals, dists, ref_gray, near_gray, ncc_gray, ncc_grad, ref_var);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_grid_sample_border_forward(int N, int channels, int height, int width, const float* image, const float* grid, float* out,
                                    void* stream) {
    if (N == 0) return GS2M_OK;
    if (N < 0 || height < 1 || width < 1 || !image || !grid || !out) return GS2M_ERR_INVALID_ARG;
    const dim3 g((N + 255) / 256), b(256);
    switch (channels) {
        case 1: grid_border_kernel<1, false><<<g, b, 0, (hipStream_t)stream>>>(N, height, width, image, grid, out, nullptr, nullptr, nullptr); break;
        case 2: grid_border_kernel<2, false><<<g, b, 0, (hipStream_t)stream>>>(N, height, width, image, grid, out, nullptr, nullptr, nullptr); break;
        case 3: grid_border_kernel<3, false><<<g, b, 0, (hipStream_t)stream>>>(N, height, width, image, grid, out, nullptr, nullptr, nullptr); break;
        case 4: grid_border_kernel<4, false><<<g, b, 0, (hipStream_t)stream>>>(N, height, width, image, grid, out, nullptr, nullptr, nullptr); break;
        default: return GS2M_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

void gs2m_mvs_set_deterministic(int on) { g_mvs_deterministic.store(on ? 1 : 0); }
int gs2m_mvs_get_deterministic(void) { return g_mvs_deterministic.load(); }

int gs2m_grid_sample_border_backward(int N, int channels, int height, int width, const float* image, const float* grid,
                                     const float* dL_dout, float* dL_dimage, float* dL_dgrid, void* stream) {
    if (N == 0) return GS2M_OK;
    if (N < 0 || height < 1 || width < 1 || !image || !grid || !dL_dout) return GS2M_ERR_INVALID_ARG;
    if (channels < 1 || channels > 4) return GS2M_ERR_UNSUPPORTED;
    const dim3 g((N + 255) / 256), b(256);
    hipStream_t s = (hipStream_t)stream;
    const bool det = g_mvs_deterministic.load() != 0 && dL_dimage != nullptr;
    DetWorkspace* ws = nullptr;
    const size_t words = (size_t)channels * height * width;
    if (det) {
        ws = det_workspace(words, s);
        if (ws == nullptr) return GS2M_ERR_ALLOC;
        if (hipMemsetAsync(ws->maxbits, 0, sizeof(uint32_t), s) != hipSuccess) return GS2M_ERR_HIP;
    }
    const int hb = det_headroom(N);
#define GS2M_GB_BWD(CH)                                                                                                                   \
    if (det) {                                                                                                                            \
        grid_border_kernel<CH, true, SCATTER_MAX><<<g, b, 0, s>>>(N, height, width, image, grid, nullptr, dL_dout, dL_dimage, dL_dgrid, ws->acc, ws->maxbits, hb);   \
        grid_border_kernel<CH, true, SCATTER_FIXED><<<g, b, 0, s>>>(N, height, width, image, grid, nullptr, dL_dout, dL_dimage, dL_dgrid, ws->acc, ws->maxbits, hb); \
    } else {                                                                                                                              \
        grid_border_kernel<CH, true><<<g, b, 0, s>>>(N, height, width, image, grid, nullptr, dL_dout, dL_dimage, dL_dgrid);               \
    }
    switch (channels) {
        case 1: GS2M_GB_BWD(1); break;
        case 2: GS2M_GB_BWD(2); break;
        case 3: GS2M_GB_BWD(3); break;
        default: GS2M_GB_BWD(4); break;
    }
#undef GS2M_GB_BWD
    if (det) det_finalize_kernel<<<(unsigned)((words + 255) / 256), 256, 0, s>>>(words, ws->acc, DetScale{ws->maxbits, hb}, dL_dimage);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

static int fill_geo(GeoConst& C, const float* A, const float* b, const float* A2, const float* b2, const float* intr_ref, const float* intr_near,
                    int width, int height, int width_n, int height_n, float occlusion) {
    if (!A || !b || !A2 || !b2 || !intr_ref || !intr_near || width < 1 || height < 1 || width_n < 2 || height_n < 2) return GS2M_ERR_INVALID_ARG;
    for (int k = 0; k < 9; k++) { C.A[k] = A[k]; C.A2[k] = A2[k]; }
    for (int k = 0; k < 3; k++) { C.b[k] = b[k]; C.b2[k] = b2[k]; }
    C.fx = intr_ref[0]; C.fy = intr_ref[1]; C.cx = intr_ref[2]; C.cy = intr_ref[3];
    C.fxn = intr_near[0]; C.fyn = intr_near[1]; C.cxn = intr_near[2]; C.cyn = intr_near[3];
    C.W = width; C.H = height; C.Wn = width_n; C.Hn = height_n; C.occlusion = occlusion;
    return GS2M_OK;
}

int gs2m_mv_geo_forward(int width, int height, int width_n, int height_n, const float* depth, const float* normal, const float* depth_n,
                        const float* normal_n, const float* A, const float* b, const float* A2, const float* b2, const float* intr_ref,
                        const float* intr_near, float occlusion, float* pixel_noise, float* angle, uint8_t* valid, void* stream) {
    if (!depth || !normal || !depth_n || !normal_n || !pixel_noise || !angle || !valid) return GS2M_ERR_INVALID_ARG;
    GeoConst C;
    const int rc = fill_geo(C, A, b, A2, b2, intr_ref, intr_near, width, height, width_n, height_n, occlusion);
    if (rc != GS2M_OK) return rc;
    const int n = width * height;
    mv_geo_kernel<false><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(C, depth, normal, depth_n, normal_n, pixel_noise, angle, valid, nullptr,
                                                                          nullptr, nullptr, nullptr, nullptr, nullptr);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_mv_geo_backward(int width, int height, int width_n, int height_n, const float* depth, const float* normal, const float* depth_n,
                         const float* normal_n, const float* A, const float* b, const float* A2, const float* b2, const float* intr_ref,
                         const float* intr_near, float occlusion, const float* dL_dnoise, const float* dL_dangle, float* dL_ddepth,
                         float* dL_dnormal, float* dL_ddepth_n, float* dL_dnormal_n, void* stream) {
    if (!depth || !normal || !depth_n || !normal_n || !dL_dnoise || !dL_dangle || !dL_ddepth || !dL_dnormal || !dL_ddepth_n || !dL_dnormal_n)
        return GS2M_ERR_INVALID_ARG;
    GeoConst C;
    const int rc = fill_geo(C, A, b, A2, b2, intr_ref, intr_near, width, height, width_n, height_n, occlusion);
    if (rc != GS2M_OK) return rc;
    const int n = width * height;
    hipStream_t s = (hipStream_t)stream;
    if (g_mvs_deterministic.load() != 0) {
        const size_t words = (size_t)4 * height_n * width_n;  // depth + three normal planes of the neighbour
        if (width_n >= (1 << 15) || height_n >= (1 << 15)) return GS2M_ERR_UNSUPPORTED;  // (the parked footprint: 15 bits per coordinate)
        DetWorkspace* ws = det_workspace(words, s, (size_t)7 * n);
        if (ws == nullptr) return GS2M_ERR_ALLOC;
        if (hipMemsetAsync(ws->maxbits, 0, sizeof(uint32_t), s) != hipSuccess) return GS2M_ERR_HIP;
        const int hb = det_headroom(n);
        mv_geo_kernel<true, SCATTER_MAX><<<(n + 255) / 256, 256, 0, s>>>(C, depth, normal, depth_n, normal_n, nullptr, nullptr, nullptr, dL_dnoise, dL_dangle,
                                                                         dL_ddepth, dL_dnormal, dL_ddepth_n, dL_dnormal_n, ws->acc, ws->maxbits, hb, ws->scratch);
        mv_geo_scatter_kernel<<<(n + 255) / 256, 256, 0, s>>>((size_t)n, width_n, (size_t)height_n * width_n, ws->scratch, ws->acc, DetScale{ws->maxbits, hb});
        const size_t plane = (size_t)height_n * width_n;
        det_finalize_kernel<<<(unsigned)((plane + 255) / 256), 256, 0, s>>>(plane, ws->acc, DetScale{ws->maxbits, hb}, dL_ddepth_n);
        det_finalize_kernel<<<(unsigned)((3 * plane + 255) / 256), 256, 0, s>>>(3 * plane, ws->acc + plane, DetScale{ws->maxbits, hb}, dL_dnormal_n);
    } else {
        mv_geo_kernel<true><<<(n + 255) / 256, 256, 0, s>>>(C, depth, normal, depth_n, normal_n, nullptr, nullptr, nullptr, dL_dnoise,
                                                            dL_dangle, dL_ddepth, dL_dnormal, dL_ddepth_n, dL_dnormal_n);
    }
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

}  // extern "C"
