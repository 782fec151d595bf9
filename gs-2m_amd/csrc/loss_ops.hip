// The per-iteration loss tail of the reference's training loop as fused kernels for gfx950 (SURVEY.md 8(f) row N1:
// "optimizer-side elementwise work"; C ABI in include/gs2m_loss.h).
//
// Around the rasterizer and the D-SSIM term the reference's geometry stage runs ~90 small PyTorch kernels per iteration
// (train.py:94-130, 223-227; utils/loss_utils.py:27-28, 72-79, 113-135; scene/gaussian_model.py:569-573): the clamp of
// the render, its L1 distance to the ground truth, the edge-aware depth-normal term (with the image-gradient weight
// recomputed from the ground truth every iteration), the plane (flattening) term and the densification statistics --
// 0.74 ms of a 2.8 ms iteration at 1M Gaussians / 1080p, every one a full pass over 6-25 MB.  Here:
//   edge_gradient     ground truth -> raw edge strength per pixel + its min / max over the interior     (1 launch)
//   image_loss fwd    render, ground truth, normal / Sobel-normal maps, edge strength -> clamped render (for the
//                     D-SSIM term) + the weighted sum  w_l1 mean|rgb - gt| + w_dn mean(edge weight * |sobel - normal|_1)
//   image_loss bwd    d loss, d rgb (from D-SSIM) -> d render, d normal map, d Sobel map                  (1 launch)
//   plane_loss        activated scales + visibility -> mean smallest scale of the visible Gaussians, and its backward
//   densification_stats   viewspace gradient norms, counts and the screen-radius maximum, in place            (1 launch)
// All of it is HBM bound: one read of every input, one write of every output.
// Sums are formed deterministically: a fixed grid, one partial per workgroup, and the last workgroup to finish (atomic
// ticket) adds the partials in index order in double precision and leaves the ticket at 0 for the next call.
#include "common.h"
#include "../../include/gs2m_loss.h"

namespace {

constexpr int LB = 256;     // threads per workgroup of the elementwise launches
// Reducing launches: one 1024-thread workgroup per CU (16 waves: the loads of a full CU in flight) and a fixed grid, so the
// partial sums do not depend on the problem size and the ticket (a same-address atomic, ~10 ns each) is taken 256 times.
constexpr int RB = 1024;
constexpr int LG = 256;
constexpr int RU = 4;      // elements per trip of a reducing kernel's grid-stride loop (their loads are issued together)
constexpr int WS_K = 4;     // partial sums per workgroup the workspace holds (the ticket word sits behind them)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroup partials -> ws[k * LG + block], then a ticket; returns true in the LAST workgroup once every partial is visible.
template <int K>
__device__ __forceinline__ bool publish_partials(const float (&v)[K], float* __restrict__ ws, uint32_t* __restrict__ ticket,
                                                 int op /* 0 sum, 1: v[0] min, v[1] max */) {
    __shared__ float s_p[K][RB / 64];
    __shared__ bool s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const float r = op == 0 ? wave_sum(v[k]) : (k == 0 ? wave_min(v[k]) : wave_max(v[k]));
        if (lane == 0) s_p[k][wave] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            float r = s_p[k][0];
#pragma unroll
            for (int w = 1; w < RB / 64; w++) r = op == 0 ? r + s_p[k][w] : (k == 0 ? fminf(r, s_p[k][w]) : fmaxf(r, s_p[k][w]));
            __hip_atomic_store(&ws[k * LG + blockIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef GS2M_TICKET_FENCED
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        // No release / acquire fence: at agent scope those write back and invalidate the XCD's whole L2 (which holds the
        // megabytes this kernel has just stored) once per workgroup.  The partials above are agent-scope atomic stores (written
        // through to memory), s_waitcnt vmcnt(0) holds the ticket back until they have been acknowledged, and the last workgroup
        // reads them with agent-scope atomic loads behind the ticket's return value.  This is NOT a release / acquire pair of
        // the HIP memory model: it is the hand-off MI355X_MICROARCH.md lists as measured-valid on gfx950 (one lane of each
        // storing workgroup: sc1 stores, s_waitcnt vmcnt(0), agent-scope atomic add; the workgroup whose add came last loads
        // with sc1 loads behind a workgroup barrier), so the build is pinned to that target below; -DGS2M_TICKET_FENCED
        // selects the portable form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "publish_partials: the unfenced ticket relies on gfx950 cache behaviour (sc1 write-through); build with -DGS2M_TICKET_FENCED elsewhere"
#endif
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        s_last = t == gridDim.x - 1;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last;
}
// In the last workgroup: the sum of partial k over the grid, in a fixed order, double precision.  Valid in thread 0.
__device__ __forceinline__ double final_sum(const float* __restrict__ ws, int k) {
    __shared__ double s_d[RB];
    double a = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += RB)
        a += (double)__hip_atomic_load(&ws[k * LG + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_d[threadIdx.x] = a;
    __syncthreads();
    for (int s = RB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_d[threadIdx.x] += s_d[threadIdx.x + s];
        __syncthreads();
    }
    const double r = s_d[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float final_minmax(const float* __restrict__ ws, int k) {
    __shared__ float s_f[RB];
    float a = k == 0 ? INFINITY : -INFINITY;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += RB) {
        const float v = __hip_atomic_load(&ws[k * LG + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = k == 0 ? fminf(a, v) : fmaxf(a, v);
    }
    s_f[threadIdx.x] = a;
    __syncthreads();
    for (int s = RB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_f[threadIdx.x] = k == 0 ? fminf(s_f[threadIdx.x], s_f[threadIdx.x + s]) : fmaxf(s_f[threadIdx.x], s_f[threadIdx.x + s]);
        __syncthreads();
    }
    const float r = s_f[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float clamp01(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }  // NaN passes, as torch.clamp
__device__ __forceinline__ float sgn(float d) { return (float)(d > 0.f) - (float)(d < 0.f); }          // torch.sgn: 0 at 0

// _get_img_grad_weight, utils/loss_utils.py:122-135, without the normalisation: per interior pixel the larger of the
// mean absolute central differences along x and along y; 0 on the one-pixel border (the reference pads).
__global__ void __launch_bounds__(RB) edge_gradient_kernel(int W, int H, const float* __restrict__ gt, float* __restrict__ g,
                                                           float* __restrict__ minmax, float* __restrict__ ws, uint32_t* __restrict__ ticket) {
    const size_t HW = (size_t)H * W;
    float v[2] = {INFINITY, -INFINITY};
    // RU pixels per trip, all their loads requested before the first use: at 1080p a thread of the fixed grid walks 8
    // pixels, and one pixel per trip is a chain of 8 memory round trips (20 us for 33 MB)
    // A workgroup owns a contiguous run of pixels (~4 rows at 1080p): the rows above and below a pixel are then mostly its
    // own (L1 / the XCD's L2); with workgroup b on pixels b RB + k LG RB they belong to workgroups on other XCDs and every
    // line comes from memory three times.
    const size_t chunk = (HW + LG - 1) / LG, cbeg = (size_t)blockIdx.x * chunk, cend = cbeg + chunk < HW ? cbeg + chunk : HW;
    for (size_t p0 = cbeg + threadIdx.x; p0 < cend; p0 += (size_t)RU * RB) {
        float t[RU][3][4];
        bool in[RU];
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const size_t p = p0 + (size_t)u * RB;
            const bool have = p < cend;
            const int y = have ? (int)((uint32_t)p / (uint32_t)W) : 0, x = have ? (int)((uint32_t)p - (uint32_t)y * (uint32_t)W) : 0;  // 32-bit: H W < 2^31 (checked by the caller)
            in[u] = have && x > 0 && x < W - 1 && y > 0 && y < H - 1;
            const size_t pc = in[u] ? p : (size_t)W + 1;   // a pixel whose four neighbours exist (W, H >= 3)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float* q = gt + c * HW + pc;
                t[u][c][0] = q[1]; t[u][c][1] = q[-1]; t[u][c][2] = q[-W]; t[u][c][3] = q[W];
            }
        }
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const size_t p = p0 + (size_t)u * RB;
            float e = 0.f;
            if (in[u]) {
                float ax[3], ay[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    ax[c] = fabsf(t[u][c][0] - t[u][c][1]);
                    ay[c] = fabsf(t[u][c][2] - t[u][c][3]);
                }
                const float gx = ((ax[0] + ax[1]) + ax[2]) / 3.0f, gy = ((ay[0] + ay[1]) + ay[2]) / 3.0f;
                e = fmaxf(gx, gy);
                v[0] = fminf(v[0], e);
                v[1] = fmaxf(v[1], e);
            }
            if (p < cend) g[p] = e;
        }
    }
    if (publish_partials<2>(v, ws, ticket, 1)) {
        const float mn = final_minmax(ws, 0), mx = final_minmax(ws, 1);
        if (threadIdx.x == 0) { minmax[0] = mn; minmax[1] = mx; }
    }
}

// (1 - normalised edge strength) clamped to [0, 1], squared (utils/loss_utils.py:117-118); 1 on the border
__device__ __forceinline__ float edge_weight(float e, float mn, float mx, bool interior) {
    const float gn = interior ? (e - mn) / (mx - mn) : 0.f;
    const float w = clamp01(1.0f - gn);
    return w * w;
}

struct ImageLossArgs {
    int W, H;
    const float *image, *gt, *normal, *sobel, *edge, *edge_minmax, *weight_map;
    float w_l1, w_dn;
    int hwc;              // image (and its gradient) are (H, W, 3) -- the shading's layout -- instead of (3, H, W)
    const uint8_t* mask;  // (H, W): pixels outside take the background colour (train.py:141-142); NULL: every pixel
    const float* bg;      // (3)
};
__device__ __forceinline__ size_t image_index(const ImageLossArgs& a, size_t HW, size_t p, int c) { return a.hwc ? 3 * p + c : c * HW + p; }

__global__ void __launch_bounds__(RB) image_loss_fwd_kernel(ImageLossArgs a, float* __restrict__ rgb, float* __restrict__ out,
                                                            float* __restrict__ ws, uint32_t* __restrict__ ticket) {
    const size_t HW = (size_t)a.H * a.W;
    const bool dn = a.normal != nullptr;
    float mn = 0.f, mx = 1.f;
    if (dn && a.edge) { mn = a.edge_minmax[0]; mx = a.edge_minmax[1]; }
    float v[2] = {0.f, 0.f};
    for (size_t p0 = (size_t)blockIdx.x * RB + threadIdx.x; p0 < HW; p0 += (size_t)RU * LG * RB) {  // RU pixels per trip: edge_gradient_kernel
        float im[RU][3], g[RU][3], so[RU][3], no[RU][3], ed[RU], wm[RU];
        bool inside[RU];
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const size_t p = p0 + (size_t)u * LG * RB, q = p < HW ? p : p0;
            inside[u] = a.mask == nullptr || a.mask[q] != 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                im[u][c] = a.image[image_index(a, HW, q, c)];
                g[u][c] = a.gt[c * HW + q];
            }
            ed[u] = wm[u] = 1.0f;
            if (dn) {
                if (a.edge) ed[u] = a.edge[q];
                if (a.weight_map) wm[u] = a.weight_map[q];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    so[u][c] = a.sobel[c * HW + q];
                    no[u][c] = a.normal[c * HW + q];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const size_t p = p0 + (size_t)u * LG * RB;
            if (p >= HW) break;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float r = inside[u] ? clamp01(im[u][c]) : a.bg[c];
                rgb[c * HW + p] = r;
                v[0] += fabsf(r - g[u][c]);
            }
            if (dn) {
                const int y = (int)((uint32_t)p / (uint32_t)a.W), x = (int)((uint32_t)p - (uint32_t)y * (uint32_t)a.W);
                float w = a.edge ? edge_weight(ed[u], mn, mx, x > 0 && x < a.W - 1 && y > 0 && y < a.H - 1) : 1.0f;
                if (a.weight_map) w *= wm[u];
                const float d0 = fabsf(so[u][0] - no[u][0]), d1 = fabsf(so[u][1] - no[u][1]), d2 = fabsf(so[u][2] - no[u][2]);
                v[1] += w * ((d0 + d1) + d2);
            }
        }
    }
    if (publish_partials<2>(v, ws, ticket, 0)) {
        const double s0 = final_sum(ws, 0), s1 = final_sum(ws, 1);
        if (threadIdx.x == 0) {
            const float l1 = (float)(s0 / (3.0 * (double)HW)), dnl = (float)(s1 / (double)HW);
            out[0] = a.w_l1 * l1 + a.w_dn * dnl;
            out[1] = l1;
            out[2] = dnl;
        }
    }
}

__global__ void __launch_bounds__(LB) image_loss_bwd_kernel(ImageLossArgs a, const float* __restrict__ g_loss, bool has_g,
                                                            const float* __restrict__ g_rgb, float* __restrict__ d_image,
                                                            float* __restrict__ d_normal, float* __restrict__ d_sobel) {
    const size_t HW = (size_t)a.H * a.W;
    const size_t p = (size_t)blockIdx.x * LB + threadIdx.x;
    if (p >= HW) return;
    const float g = has_g ? g_loss[0] : 0.f;
    const float k1 = g * a.w_l1 / (3.0f * (float)HW);
    const bool inside = a.mask == nullptr || a.mask[p] != 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float x = a.image[image_index(a, HW, p, c)];
        const float up = (g_rgb ? g_rgb[c * HW + p] : 0.f) + k1 * sgn(clamp01(x) - a.gt[c * HW + p]);
        // clamp's backward: the gradient passes inside [min, max]; masked-out pixels show the background: no gradient
        d_image[image_index(a, HW, p, c)] = (inside && x >= 0.f && x <= 1.f) ? up : 0.f;
    }
    if (a.normal != nullptr) {
        const int y = (int)((uint32_t)p / (uint32_t)a.W), xx = (int)((uint32_t)p - (uint32_t)y * (uint32_t)a.W);
        float w = a.edge ? edge_weight(a.edge[p], a.edge_minmax[0], a.edge_minmax[1], xx > 0 && xx < a.W - 1 && y > 0 && y < a.H - 1) : 1.0f;
        if (a.weight_map) w *= a.weight_map[p];
        const float k2 = g * a.w_dn / (float)HW * w;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float s = k2 * sgn(a.sobel[c * HW + p] - a.normal[c * HW + p]);
            d_sobel[c * HW + p] = s;
            d_normal[c * HW + p] = -s;
        }
    }
}

// plane_loss, utils/loss_utils.py:72-79: the mean over the visible Gaussians of their smallest scale
// (raw: `scaling` holds the log-scales, scene/gaussian_model.py:113-114 -- exp is monotone, so min exp(s) = exp(min s))
__global__ void __launch_bounds__(RB) plane_loss_fwd_kernel(int P, const float* __restrict__ scaling, int raw, const uint8_t* __restrict__ vis,
                                                            float weight, float* __restrict__ out, float* __restrict__ ws, uint32_t* __restrict__ ticket) {
    float v[2] = {0.f, 0.f};
    // four elements per trip, every load requested before the first use and none behind the visibility test: the loop is
    // a chain of memory round trips otherwise (two per element at 1 M Gaussians on 262144 threads)
    constexpr int U = 4;
    for (int i0 = blockIdx.x * RB + threadIdx.x; i0 < P; i0 += U * LG * RB) {
        uint8_t seen[U];
        float sc[U][3];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * LG * RB;
            const int j = i < P ? i : i0;
            seen[u] = i < P ? vis[j] : (uint8_t)0;
#pragma unroll
            for (int c = 0; c < 3; c++) sc[u][c] = scaling[3 * (size_t)j + c];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {  // same order of additions as the one-element loop
            if (seen[u]) {
                const float m = fminf(fminf(sc[u][0], sc[u][1]), sc[u][2]);
                v[0] += raw ? expf(m) : m;
                v[1] += 1.0f;
            }
        }
    }
    if (publish_partials<2>(v, ws, ticket, 0)) {
        const double s = final_sum(ws, 0), n = final_sum(ws, 1);
        if (threadIdx.x == 0) {
            out[0] = weight * (float)(s / (n > 1.0 ? n : 1.0));
            out[1] = (float)n;
        }
    }
}
__global__ void __launch_bounds__(LB) plane_loss_bwd_kernel(int P, const float* __restrict__ scaling, int raw, const uint8_t* __restrict__ vis,
                                                            float weight, const float* __restrict__ out, const float* __restrict__ g,
                                                            float* __restrict__ d_scaling) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= P) return;
    const float n = out[1];
    float k = vis[i] ? (g[0] * weight) / (n > 1.0f ? n : 1.0f) : 0.f;
    const float s0 = scaling[3 * i], s1 = scaling[3 * i + 1], s2 = scaling[3 * i + 2];
    int m = 0;  // the first index of the minimum takes the gradient
    float sm = s0;
    if (s1 < sm) { sm = s1; m = 1; }
    if (s2 < sm) { sm = s2; m = 2; }
    if (raw) k *= expf(sm);  // d exp(s) / d s
    d_scaling[3 * i] = m == 0 ? k : 0.f;
    d_scaling[3 * i + 1] = m == 1 ? k : 0.f;
    d_scaling[3 * i + 2] = m == 2 ? k : 0.f;
}

// add_densification_stats, scene/gaussian_model.py:569-573, and the max_radii2D update of train.py:223-225
__global__ void __launch_bounds__(LB) densification_stats_kernel(int P, const float4* __restrict__ vg, const uint8_t* __restrict__ vis,
                                                                 const int* __restrict__ observe, const int* __restrict__ radii,
                                                                 float* __restrict__ accum, float* __restrict__ accum_abs,
                                                                 float* __restrict__ denom, float* __restrict__ max_radii) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= P || !vis[i]) return;
    const float4 g = vg[i];
    accum[i] += sqrtf(g.x * g.x + g.y * g.y);
    accum_abs[i] += sqrtf(g.z * g.z + g.w * g.w);
    denom[i] += 1.0f;
    if (max_radii != nullptr && observe[i] > 0) max_radii[i] = fmaxf(max_radii[i], (float)radii[i]);
}

// tv_loss, utils/loss_utils.py:536-557: edge-aware total variation of pred (C, H, W).  Neighbour differences along y and
// along x (absolute, or squared with norm1 = 0), damped by exp(-channel-mean |difference of the ground truth|) and, with a
// weight map, by the mean of the two pixels' weights; the mean over all vertical pairs plus the mean over all horizontal pairs.
struct TvArgs {
    int W, H, C, norm1;
    const float *gt, *pred, *wm;
    float weight;  // the term's multiplier in the loss (lambda): out = weight * tv, the backward scales by it
};
// damping and weight of the pair (p, p + stride): stride = W (vertical) or 1 (horizontal)
__device__ __forceinline__ float tv_pair_weight(const TvArgs& a, size_t HW, size_t p, size_t stride) {
    const float e0 = fabsf(a.gt[p + stride] - a.gt[p]), e1 = fabsf(a.gt[HW + p + stride] - a.gt[HW + p]),
                e2 = fabsf(a.gt[2 * HW + p + stride] - a.gt[2 * HW + p]);
    float w = expf(-(((e0 + e1) + e2) / 3.0f));
    if (a.wm) w *= (a.wm[p + stride] + a.wm[p]) / 2.0f;
    return w;
}
__global__ void __launch_bounds__(RB) tv_loss_fwd_kernel(TvArgs a, float* __restrict__ out, float* __restrict__ ws, uint32_t* __restrict__ ticket) {
    const size_t HW = (size_t)a.H * a.W;
    float v[2] = {0.f, 0.f};
    // TU pixels per trip; the loads of a trip carry no test (a pair that does not exist reads the pixel itself and is not
    // added): edge_gradient_kernel.  Channels beyond 3 (none in the reference's calls) take the one-pixel path.
    constexpr int TU = 2;
    if (a.C <= 3) {
        // a contiguous run of pixels per workgroup (edge_gradient_kernel): the row below is mostly its own
        const size_t chunk = (HW + LG - 1) / LG, cbeg = (size_t)blockIdx.x * chunk, cend = cbeg + chunk < HW ? cbeg + chunk : HW;
        for (size_t p0 = cbeg + threadIdx.x; p0 < cend; p0 += (size_t)TU * RB) {
            float gq[TU][3][3], pq[TU][3][3], wq[TU][3];
            bool down[TU], right[TU];
#pragma unroll
            for (int u = 0; u < TU; u++) {
                const size_t p = p0 + (size_t)u * RB, q = p < cend ? p : p0;
                const int y = (int)((uint32_t)q / (uint32_t)a.W), x = (int)((uint32_t)q - (uint32_t)y * (uint32_t)a.W);
                down[u] = p < cend && y < a.H - 1;
                right[u] = p < cend && x < a.W - 1;
                const size_t qd = down[u] ? q + a.W : q, qr = right[u] ? q + 1 : q;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    gq[u][c][0] = a.gt[c * HW + q]; gq[u][c][1] = a.gt[c * HW + qd]; gq[u][c][2] = a.gt[c * HW + qr];
                    if (c < a.C) { pq[u][c][0] = a.pred[c * HW + q]; pq[u][c][1] = a.pred[c * HW + qd]; pq[u][c][2] = a.pred[c * HW + qr]; }
                }
                if (a.wm) { wq[u][0] = a.wm[q]; wq[u][1] = a.wm[qd]; wq[u][2] = a.wm[qr]; }
            }
#pragma unroll
            for (int u = 0; u < TU; u++) {
#pragma unroll
                for (int k = 1; k <= 2; k++) {  // k = 1: the pair (p, p + W); k = 2: (p, p + 1)
                    if (k == 1 ? down[u] : right[u]) {
                        const float e0 = fabsf(gq[u][0][k] - gq[u][0][0]), e1 = fabsf(gq[u][1][k] - gq[u][1][0]), e2 = fabsf(gq[u][2][k] - gq[u][2][0]);
                        float w = expf(-(((e0 + e1) + e2) / 3.0f));
                        if (a.wm) w *= (wq[u][k] + wq[u][0]) / 2.0f;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            if (c < a.C) {
                                const float d = pq[u][c][k] - pq[u][c][0];
                                v[k - 1] += (a.norm1 ? fabsf(d) : d * d) * w;
                            }
                        }
                    }
                }
            }
        }
    } else {
        for (size_t p = (size_t)blockIdx.x * RB + threadIdx.x; p < HW; p += (size_t)LG * RB) {
            const int y = (int)((uint32_t)p / (uint32_t)a.W), x = (int)((uint32_t)p - (uint32_t)y * (uint32_t)a.W);
            if (y < a.H - 1) {
                const float w = tv_pair_weight(a, HW, p, a.W);
                for (int c = 0; c < a.C; c++) {
                    const float d = a.pred[c * HW + p + a.W] - a.pred[c * HW + p];
                    v[0] += (a.norm1 ? fabsf(d) : d * d) * w;
                }
            }
            if (x < a.W - 1) {
                const float w = tv_pair_weight(a, HW, p, 1);
                for (int c = 0; c < a.C; c++) {
                    const float d = a.pred[c * HW + p + 1] - a.pred[c * HW + p];
                    v[1] += (a.norm1 ? fabsf(d) : d * d) * w;
                }
            }
        }
    }
    if (publish_partials<2>(v, ws, ticket, 0)) {
        const double sh = final_sum(ws, 0), sw = final_sum(ws, 1);
        if (threadIdx.x == 0)
            out[0] = a.weight * ((float)(sh / ((double)a.C * (a.H - 1) * a.W)) + (float)(sw / ((double)a.C * a.H * (a.W - 1))));
    }
}
// gather form: pixel p is the lower end of its own vertical / horizontal pair and the upper end of the pairs that start
// one row up / one column left
__global__ void __launch_bounds__(LB) tv_loss_bwd_kernel(TvArgs a, const float* __restrict__ g_loss, float* __restrict__ d_pred) {
    const size_t HW = (size_t)a.H * a.W;
    // workgroups are dealt to the 8 XCDs round robin: block b works on pixel block (b % 8) * ceil(n / 8) + b / 8, so that an
    // XCD's L2 sees a contiguous eighth of the image and the rows above and below a pixel hit it
    // (the grid is a multiple of 8)
    const uint32_t per = gridDim.x >> 3, lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const size_t p = (size_t)lb * LB + threadIdx.x;
    if (p >= HW) return;
    const int y = (int)((uint32_t)p / (uint32_t)a.W), x = (int)((uint32_t)p - (uint32_t)y * (uint32_t)a.W);
    const float g = g_loss[0] * a.weight;
    const float kh = g / (float)((double)a.C * (a.H - 1) * a.W), kw = g / (float)((double)a.C * a.H * (a.W - 1));
    const float wd = y < a.H - 1 ? kh * tv_pair_weight(a, HW, p, a.W) : 0.f;       // pair (p, p + W)
    const float wu = y > 0 ? kh * tv_pair_weight(a, HW, p - a.W, a.W) : 0.f;      // pair (p - W, p)
    const float wr = x < a.W - 1 ? kw * tv_pair_weight(a, HW, p, 1) : 0.f;         // pair (p, p + 1)
    const float wl = x > 0 ? kw * tv_pair_weight(a, HW, p - 1, 1) : 0.f;           // pair (p - 1, p)
    for (int c = 0; c < a.C; c++) {
        const float* q = a.pred + c * HW + p;
        const float v0 = q[0];
        auto dd = [&](float d) { return a.norm1 ? sgn(d) : 2.0f * d; };
        float r = 0.f;
        if (y < a.H - 1) r -= wd * dd(q[a.W] - v0);
        if (y > 0) r += wu * dd(v0 - q[-a.W]);
        if (x < a.W - 1) r -= wr * dd(q[1] - v0);
        if (x > 0) r += wl * dd(v0 - q[-1]);
        d_pred[c * HW + p] = r;
    }
}

// The photometric part of multi_view_loss around patch_ncc (utils/loss_utils.py:293-300, 345-349), without the framework's gathers and
// its reduction chain.  (a) mv_take: what the sampled pixels `idx` (distinct indices into the frame) hand the NCC kernel -- pixel
// coordinates, the plane normal and distance at the pixel, the sample's weight -- in one launch instead of a (3,H,W)->(HW,3) copy and four
// index kernels; the backward scatters the normals' and distances' gradients into zero-filled maps (no sort: the indices are distinct).
// (b) ncc_tail: mask = ncc < 0.9, loss = sum(ncc w mask) / max(#mask, 1), one launch each way instead of ~8.
struct MvTakeArgs {
    int N, W;
    size_t HW;
    const long long* idx;
};
__global__ void __launch_bounds__(LB) mv_take_fwd_kernel(MvTakeArgs a, const float* __restrict__ normal, const float* __restrict__ dist, const float* __restrict__ w,
                                                         float* __restrict__ pixels, float* __restrict__ normals, float* __restrict__ dists, float* __restrict__ wout) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= a.N) return;
    const size_t p = (size_t)a.idx[i];
    const uint32_t y = (uint32_t)(p / (size_t)a.W), x = (uint32_t)(p - (size_t)y * (size_t)a.W);
    pixels[2 * i] = (float)x;
    pixels[2 * i + 1] = (float)y;
#pragma unroll
    for (int c = 0; c < 3; c++) normals[3 * i + c] = normal[c * a.HW + p];
    dists[i] = dist[p];
    wout[i] = w ? w[p] : 1.0f;
}
__global__ void __launch_bounds__(LB) mv_take_bwd_kernel(MvTakeArgs a, const float* __restrict__ d_normals, const float* __restrict__ d_dists,
                                                         float* __restrict__ d_normal /* (3, HW), zero-filled */, float* __restrict__ d_dist /* (HW), zero-filled */) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= a.N) return;
    const size_t p = (size_t)a.idx[i];
#pragma unroll
    for (int c = 0; c < 3; c++) d_normal[c * a.HW + p] = d_normals[3 * i + c];
    d_dist[p] = d_dists[i];
}
__global__ void __launch_bounds__(RB) ncc_tail_fwd_kernel(int N, const float* __restrict__ ncc, const float* __restrict__ w, float* __restrict__ out,
                                                          float* __restrict__ ws, uint32_t* __restrict__ ticket) {
    float v[2] = {0.f, 0.f};
    for (int i = blockIdx.x * RB + threadIdx.x; i < N; i += LG * RB) {
        const float c = ncc[i];
        const bool m = c < 0.9f;
        v[0] += m ? c * w[i] : 0.f;
        v[1] += m ? 1.0f : 0.f;
    }
    if (publish_partials<2>(v, ws, ticket, 0)) {
        const double s = final_sum(ws, 0), n = final_sum(ws, 1);
        if (threadIdx.x == 0) {
            out[0] = (float)(s / (n > 1.0 ? n : 1.0));
            out[1] = (float)n;
        }
    }
}
__global__ void __launch_bounds__(LB) ncc_tail_bwd_kernel(int N, const float* __restrict__ ncc, const float* __restrict__ w, const float* __restrict__ out,
                                                          const float* __restrict__ g_loss, float* __restrict__ d_ncc) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= N) return;
    const float n = out[1];
    d_ncc[i] = ncc[i] < 0.9f ? g_loss[0] * w[i] / (n > 1.0f ? n : 1.0f) : 0.f;
}

// The geometric part of multi_view_loss, utils/loss_utils.py:277-291, from the per-pixel quantities of mv_geo (csrc/mvs.hip):
//   pixel_valid = valid & (noise < 1),  angle_valid = valid & (angle < angle_threshold),  geo_w = exp(-decay noise) on
//   pixel_valid (detached),  loss = weight (sum geo_w noise / #pixel_valid + sum geo_w factor angle [angle_valid] / #angle_valid)
// plus what the photometric part samples from: the mask itself and w_ncc = exp(-noise) on it.  ~40 framework kernels over
// 2 M pixels (0.3 ms) become one launch each way.
struct MvGeoArgs {
    int N;
    const float *noise, *angle;
    const uint8_t* valid;
    float angle_threshold, decay, factor, weight;
};
__global__ void __launch_bounds__(RB) mv_geo_loss_fwd_kernel(MvGeoArgs a, float* __restrict__ out, uint8_t* __restrict__ pixel_valid,
                                                             float* __restrict__ w_ncc, float* __restrict__ ws, uint32_t* __restrict__ ticket) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * RB + threadIdx.x; i < a.N; i += LG * RB) {
        const float nz = a.noise[i], an = a.angle[i];
        const bool ok = a.valid[i] != 0;
        const bool pv = ok && nz < 1.0f, av = ok && an < a.angle_threshold;
        const float gw = pv ? expf(-nz * a.decay) : 0.f;
        v[0] += pv ? gw * nz : 0.f;  // the reference indexes [pixel_valid] (loss_utils.py:286): an invalid pixel's inf / NaN noise never enters the sum
        v[1] += pv ? 1.0f : 0.f;
        v[2] += av ? gw * (a.factor * an) : 0.f;
        v[3] += av ? 1.0f : 0.f;
        pixel_valid[i] = pv ? 1 : 0;
        w_ncc[i] = pv ? expf(-nz) : 0.f;
    }
    if (publish_partials<4>(v, ws, ticket, 0)) {
        const double s0 = final_sum(ws, 0), n0 = final_sum(ws, 1), s1 = final_sum(ws, 2), n1 = final_sum(ws, 3);
        if (threadIdx.x == 0) {
            out[0] = a.weight * ((float)(s0 / (n0 > 1.0 ? n0 : 1.0)) + (float)(s1 / (n1 > 1.0 ? n1 : 1.0)));
            out[1] = (float)n0;
            out[2] = (float)n1;
        }
    }
}
__global__ void __launch_bounds__(LB) mv_geo_loss_bwd_kernel(MvGeoArgs a, const float* __restrict__ out, const float* __restrict__ g_loss,
                                                             float* __restrict__ d_noise, float* __restrict__ d_angle) {
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= a.N) return;
    const float nz = a.noise[i], an = a.angle[i];
    const bool ok = a.valid[i] != 0;
    const bool pv = ok && nz < 1.0f, av = ok && an < a.angle_threshold;
    const float gw = pv ? expf(-nz * a.decay) : 0.f;  // detached in the loss: a weight, not a function of the noise
    const float g = g_loss[0] * a.weight, n0 = out[1], n1 = out[2];
    d_noise[i] = g * gw / (n0 > 1.0f ? n0 : 1.0f);
    d_angle[i] = av ? g * gw * a.factor / (n1 > 1.0f ? n1 : 1.0f) : 0.f;
}

// out[0] = a + b * mean(x): the mean of a map as a loss term (fused_ssim's `.mean()`, the D-SSIM term lambda (1 - mean))
__global__ void __launch_bounds__(RB) affine_mean_kernel(size_t n, const float* __restrict__ x, float a, float b, float* __restrict__ out,
                                                         float* __restrict__ ws, uint32_t* __restrict__ ticket) {
    float v[1] = {0.f};
    const size_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < n4; i += (size_t)LG * RB) {
        const float4 q = x4[i];
        v[0] += (q.x + q.y) + (q.z + q.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) v[0] += x[(n4 << 2) + threadIdx.x];
    if (publish_partials<1>(v, ws, ticket, 0)) {
        const double s = final_sum(ws, 0);
        if (threadIdx.x == 0) out[0] = a + b * (float)(s / (double)n);
    }
}

// ---------------------------------------------------------------- random subset of the set elements of a mask (gs2m_mvs.random_subset)
// The reference draws its patch samples with idx[torch.randperm(idx.numel())[:k]] (utils/loss_utils.py:283-286).  gs2m_mvs.random_subset
// (round 5) does without the sort of one random key per valid pixel -- thinning to k + 4 sqrt(k) expected survivors, then the ~1 % in excess
// removed one per stratum -- but as ~27 framework operators of a few microseconds each (`nonzero` alone is seven launches).  The same two
// steps as four launches: counter-based uniform numbers (a 64-bit mix of seed and element index: no generator state, no ordering between
// threads), an ORDERED compaction (the samples stay in pixel order: what keeps the NCC kernels' gathers coherent).
__device__ __forceinline__ unsigned long long subset_mix(unsigned long long x) {  // splitmix64's finalizer
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ bool subset_survives(const unsigned char* __restrict__ mask, int i, int n, unsigned long long seed, float p) {
    if (i >= n || mask[i] == 0) return false;
    const float u = (float)(subset_mix(seed ^ ((unsigned long long)i * 0xD1342543DE82EF95ull)) >> 40) * (1.0f / 16777216.0f);
    return u < p;
}
__device__ __forceinline__ float subset_p(int k, int set) {  // (k + 4 sqrt(k)) / count, at most 1: as gs2m_mvs.random_subset computes it
    return fminf(((float)k + 4.0f * sqrtf((float)k)) / (float)max(set, 1), 1.0f);
}
constexpr int SUB_B = 1024;
__global__ void __launch_bounds__(SUB_B) subset_count_kernel(int n, const unsigned char* __restrict__ mask, int* __restrict__ counts) {
    __shared__ int s_w[SUB_B / 64];
    const int i = blockIdx.x * SUB_B + threadIdx.x;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(i < n && mask[i] != 0);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(b);
    gs2m_sync();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SUB_B / 64; w++) t += s_w[w];
        if (t) atomicAdd(&counts[0], t);  // (integers: the order does not matter)
    }
}
__global__ void __launch_bounds__(SUB_B) subset_block_kernel(int n, const unsigned char* __restrict__ mask, int k, unsigned long long seed,
                                                              const int* __restrict__ counts, int* __restrict__ block_counts) {
    __shared__ int s_w[SUB_B / 64];
    const int i = blockIdx.x * SUB_B + threadIdx.x;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(subset_survives(mask, i, n, seed, subset_p(k, counts[0])));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(b);
    gs2m_sync();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SUB_B / 64; w++) t += s_w[w];
        block_counts[blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(SUB_B) subset_write_kernel(int n, const unsigned char* __restrict__ mask, int k, unsigned long long seed, int* __restrict__ counts,
                                                              const int* __restrict__ block_counts, long long* __restrict__ idx, int cap) {
    __shared__ int s_w[SUB_B / 64], s_part[SUB_B / 64];
    int before = 0;  // survivors in the blocks in front (chain-free: every block adds them up itself)
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SUB_B) before += block_counts[b];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
    const int i = blockIdx.x * SUB_B + threadIdx.x;
    const bool keep = subset_survives(mask, i, n, seed, subset_p(k, counts[0]));
    const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_w[wave] = __popcll(b); s_part[wave] = before; }
    gs2m_sync();
    int base = 0, off = 0, tot = 0;
    for (int w = 0; w < SUB_B / 64; w++) {
        base += s_part[w];
        if (w < wave) off += s_w[w];
        tot += s_w[w];
    }
    const int pos = base + off + __popcll(b & ((1ull << lane) - 1ull));
    if (keep && pos < cap) idx[pos] = i;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) counts[1] = base + tot;  // all survivors (may exceed cap: the caller looks)
}
// out[j] = idx[j + #{i < e : removed_i - i <= j}], removed_i = min(m - 1, floor((i + U_i) m / e)): the j-th survivor that is kept
__global__ void __launch_bounds__(LB) subset_remove_kernel(int m, int k, unsigned long long seed, const long long* __restrict__ idx, long long* __restrict__ out) {
    const int j = blockIdx.x * LB + threadIdx.x;
    if (j >= k) return;
    const int e = m - k;
    const double step = (double)m / (double)e;
    auto key = [&](int i) {  // removed_i - i (non-decreasing in i)
        const double u = (double)(subset_mix(seed ^ ((unsigned long long)i * 0xD1342543DE82EF95ull)) >> 11) * (1.0 / 9007199254740992.0);
        long long r = (long long)(((double)i + u) * step);
        if (r > m - 1) r = m - 1;
        return r - i;
    };
    int lo = 0, hi = e;  // first i whose key exceeds j = the number of keys <= j (searchsorted right=True)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key(mid) <= (long long)j) lo = mid + 1; else hi = mid;
    }
    out[j] = idx[j + lo];
}

inline int launched() { return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP; }
inline bool too_large(int W, int H) { return (long long)W * H >= (1ll << 31); }  // the kernels split a pixel index with 32-bit arithmetic

}  // namespace

extern "C" {

int gs2m_loss_workspace_bytes(void) { return (int)(WS_K * LG * sizeof(float) + 64); }

int gs2m_edge_gradient(int W, int H, const float* gt, float* edge, float* edge_minmax, void* workspace, void* stream) {
    if (W < 3 || H < 3 || !gt || !edge || !edge_minmax || !workspace) return GS2M_ERR_INVALID_ARG;
    if (too_large(W, H)) return GS2M_ERR_UNSUPPORTED;
    float* ws = (float*)workspace;
    edge_gradient_kernel<<<LG, RB, 0, (hipStream_t)stream>>>(W, H, gt, edge, edge_minmax, ws, (uint32_t*)(ws + WS_K * LG));
    return launched();
}

int gs2m_image_loss_forward(int W, int H, const float* image, int image_hwc, const unsigned char* mask, const float* background,
                            const float* gt, const float* normal_map, const float* sobel_map,
                            const float* edge, const float* edge_minmax, const float* weight_map, float w_l1, float w_dn,
                            float* rgb, float* out, void* workspace, void* stream) {
    if (W < 1 || H < 1 || !image || !gt || !rgb || !out || !workspace || (mask && !background)) return GS2M_ERR_INVALID_ARG;
    if (too_large(W, H)) return GS2M_ERR_UNSUPPORTED;
    if ((normal_map == nullptr) != (sobel_map == nullptr) || (edge == nullptr) != (edge_minmax == nullptr)) return GS2M_ERR_INVALID_ARG;
    float* ws = (float*)workspace;
    // (no NULL for the wave-uniform scalars: the compiler hoists scalar loads such as bg[c] above the test that guards them;
    // gt is valid memory and the values are not used)
    const ImageLossArgs a = {W, H, image, gt, normal_map, sobel_map, edge, edge ? edge_minmax : gt, weight_map, w_l1, w_dn, image_hwc, mask, background ? background : gt};
    image_loss_fwd_kernel<<<LG, RB, 0, (hipStream_t)stream>>>(a, rgb, out, ws, (uint32_t*)(ws + WS_K * LG));
    return launched();
}

int gs2m_image_loss_backward(int W, int H, const float* image, int image_hwc, const unsigned char* mask, const float* gt,
                             const float* normal_map, const float* sobel_map,
                             const float* edge, const float* edge_minmax, const float* weight_map, float w_l1, float w_dn,
                             const float* g_loss, const float* g_rgb, float* d_image, float* d_normal_map, float* d_sobel_map,
                             void* stream) {
    if (W < 1 || H < 1 || !image || !gt || !d_image) return GS2M_ERR_INVALID_ARG;
    if (too_large(W, H)) return GS2M_ERR_UNSUPPORTED;
    if ((normal_map == nullptr) != (sobel_map == nullptr) || (edge == nullptr) != (edge_minmax == nullptr)) return GS2M_ERR_INVALID_ARG;
    if (normal_map && (!d_normal_map || !d_sobel_map)) return GS2M_ERR_INVALID_ARG;
    const ImageLossArgs a = {W, H, image, gt, normal_map, sobel_map, edge, edge ? edge_minmax : gt, weight_map, w_l1, w_dn, image_hwc, mask, gt};
    const size_t HW = (size_t)W * H;
    image_loss_bwd_kernel<<<(unsigned)((HW + LB - 1) / LB), LB, 0, (hipStream_t)stream>>>(a, g_loss ? g_loss : gt, g_loss != nullptr, g_rgb, d_image,
                                                                                          d_normal_map, d_sobel_map);
    return launched();
}

int gs2m_tv_loss_forward(int W, int H, int C, const float* gt, const float* pred, const float* weight_map, int norm1, float weight,
                         float* out, void* workspace, void* stream) {
    if (W < 2 || H < 2 || C < 1 || !gt || !pred || !out || !workspace) return GS2M_ERR_INVALID_ARG;
    if (too_large(W, H)) return GS2M_ERR_UNSUPPORTED;
    float* ws = (float*)workspace;
    const TvArgs a = {W, H, C, norm1, gt, pred, weight_map, weight};
    tv_loss_fwd_kernel<<<LG, RB, 0, (hipStream_t)stream>>>(a, out, ws, (uint32_t*)(ws + WS_K * LG));
    return launched();
}

int gs2m_tv_loss_backward(int W, int H, int C, const float* gt, const float* pred, const float* weight_map, int norm1, float weight,
                          const float* g_loss, float* d_pred, void* stream) {
    if (W < 2 || H < 2 || C < 1 || !gt || !pred || !g_loss || !d_pred) return GS2M_ERR_INVALID_ARG;
    if (too_large(W, H)) return GS2M_ERR_UNSUPPORTED;
    const TvArgs a = {W, H, C, norm1, gt, pred, weight_map, weight};
    const size_t HW = (size_t)W * H;
    tv_loss_bwd_kernel<<<(unsigned)(((HW + LB - 1) / LB + 7) / 8 * 8), LB, 0, (hipStream_t)stream>>>(a, g_loss, d_pred);
    return launched();
}

int gs2m_mv_geo_loss_forward(int n, const float* noise, const float* angle, const unsigned char* valid, float angle_threshold, float decay,
                             float factor, float weight, float* out, unsigned char* pixel_valid, float* w_ncc, void* workspace, void* stream) {
    if (n < 1 || !noise || !angle || !valid || !out || !pixel_valid || !w_ncc || !workspace) return GS2M_ERR_INVALID_ARG;
    float* ws = (float*)workspace;
    const MvGeoArgs a = {n, noise, angle, valid, angle_threshold, decay, factor, weight};
    mv_geo_loss_fwd_kernel<<<LG, RB, 0, (hipStream_t)stream>>>(a, out, pixel_valid, w_ncc, ws, (uint32_t*)(ws + WS_K * LG));
    return launched();
}

int gs2m_mv_geo_loss_backward(int n, const float* noise, const float* angle, const unsigned char* valid, float angle_threshold, float decay,
                              float factor, float weight, const float* out, const float* g_loss, float* d_noise, float* d_angle,
                              void* stream) {
    if (n < 1 || !noise || !angle || !valid || !out || !g_loss || !d_noise || !d_angle) return GS2M_ERR_INVALID_ARG;
    const MvGeoArgs a = {n, noise, angle, valid, angle_threshold, decay, factor, weight};
    mv_geo_loss_bwd_kernel<<<(n + LB - 1) / LB, LB, 0, (hipStream_t)stream>>>(a, out, g_loss, d_noise, d_angle);
    return launched();
}

int gs2m_mv_take_forward(int n, const long long* idx, int width, int height, const float* normal_map, const float* dist_map, const float* w_map,
                         float* pixels, float* normals, float* dists, float* w, void* stream) {
    if (n < 1 || width < 1 || height < 1 || !idx || !normal_map || !dist_map || !pixels || !normals || !dists || !w) return GS2M_ERR_INVALID_ARG;
    const MvTakeArgs a = {n, width, (size_t)width * height, idx};
    mv_take_fwd_kernel<<<(n + LB - 1) / LB, LB, 0, (hipStream_t)stream>>>(a, normal_map, dist_map, w_map, pixels, normals, dists, w);
    return launched();
}

int gs2m_mv_take_backward(int n, const long long* idx, int width, int height, const float* d_normals, const float* d_dists, float* d_normal_map,
                          float* d_dist_map, void* stream) {
    if (n < 1 || width < 1 || height < 1 || !idx || !d_normals || !d_dists || !d_normal_map || !d_dist_map) return GS2M_ERR_INVALID_ARG;
    const MvTakeArgs a = {n, width, (size_t)width * height, idx};
    mv_take_bwd_kernel<<<(n + LB - 1) / LB, LB, 0, (hipStream_t)stream>>>(a, d_normals, d_dists, d_normal_map, d_dist_map);
    return launched();
}

int gs2m_subset_thin(int n, const unsigned char* mask, int k, unsigned long long seed, long long* idx, int cap, int* counts, int* block_counts, void* stream) {
    if (n < 1 || k < 1 || cap < 1 || !mask || !idx || !counts || !block_counts) return GS2M_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (n + SUB_B - 1) / SUB_B;
    if (gs2m_zero_async(counts, 2 * sizeof(int), s) != hipSuccess) return GS2M_ERR_HIP;
    subset_count_kernel<<<blocks, SUB_B, 0, s>>>(n, mask, counts);
    subset_block_kernel<<<blocks, SUB_B, 0, s>>>(n, mask, k, seed, counts, block_counts);
    subset_write_kernel<<<blocks, SUB_B, 0, s>>>(n, mask, k, seed, counts, block_counts, idx, cap);
    return launched();
}

int gs2m_subset_remove(int m, int k, unsigned long long seed, const long long* idx, long long* out, void* stream) {
    if (k < 1 || m <= k || !idx || !out) return GS2M_ERR_INVALID_ARG;
    subset_remove_kernel<<<(k + LB - 1) / LB, LB, 0, (hipStream_t)stream>>>(m, k, seed, idx, out);
    return launched();
}

int gs2m_ncc_tail_forward(int n, const float* ncc, const float* w, float* out, void* workspace, void* stream) {
    if (n < 1 || !ncc || !w || !out || !workspace) return GS2M_ERR_INVALID_ARG;
    float* ws = (float*)workspace;
    ncc_tail_fwd_kernel<<<LG, RB, 0, (hipStream_t)stream>>>(n, ncc, w, out, ws, (uint32_t*)(ws + WS_K * LG));
    return launched();
}

int gs2m_ncc_tail_backward(int n, const float* ncc, const float* w, const float* out, const float* g_loss, float* d_ncc, void* stream) {
    if (n < 1 || !ncc || !w || !out || !g_loss || !d_ncc) return GS2M_ERR_INVALID_ARG;
    ncc_tail_bwd_kernel<<<(n + LB - 1) / LB, LB, 0, (hipStream_t)stream>>>(n, ncc, w, out, g_loss, d_ncc);
    return launched();
}

int gs2m_affine_mean(long long n, const float* x, float a, float b, float* out, void* workspace, void* stream) {
    if (n <= 0 || !x || !out || !workspace || ((uintptr_t)x & 15)) return GS2M_ERR_INVALID_ARG;
    float* ws = (float*)workspace;
    affine_mean_kernel<<<LG, RB, 0, (hipStream_t)stream>>>((size_t)n, x, a, b, out, ws, (uint32_t*)(ws + WS_K * LG));
    return launched();
}

int gs2m_plane_loss_forward(int P, const float* scaling, int raw, const unsigned char* visible, float weight, float* out, void* workspace,
                            void* stream) {
    if (P < 0 || !out || !workspace || (P > 0 && (!scaling || !visible))) return GS2M_ERR_INVALID_ARG;
    float* ws = (float*)workspace;
    plane_loss_fwd_kernel<<<LG, RB, 0, (hipStream_t)stream>>>(P, scaling, raw, visible, weight, out, ws, (uint32_t*)(ws + WS_K * LG));
    return launched();
}

int gs2m_plane_loss_backward(int P, const float* scaling, int raw, const unsigned char* visible, float weight, const float* out, const float* g_loss,
                             float* d_scaling, void* stream) {
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if (P == 0) return GS2M_OK;
    if (!scaling || !visible || !out || !g_loss || !d_scaling) return GS2M_ERR_INVALID_ARG;
    plane_loss_bwd_kernel<<<(P + LB - 1) / LB, LB, 0, (hipStream_t)stream>>>(P, scaling, raw, visible, weight, out, g_loss, d_scaling);
    return launched();
}

int gs2m_densification_stats(int P, const float* viewspace_grad, const unsigned char* visible, const int* observe, const int* radii,
                             float* grad_accum, float* grad_accum_abs, float* denom, float* max_radii, void* stream) {
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if (P == 0) return GS2M_OK;
    if (!viewspace_grad || !visible || !grad_accum || !grad_accum_abs || !denom) return GS2M_ERR_INVALID_ARG;
    if (max_radii && (!observe || !radii)) return GS2M_ERR_INVALID_ARG;
    densification_stats_kernel<<<(P + LB - 1) / LB, LB, 0, (hipStream_t)stream>>>(
        P, (const float4*)viewspace_grad, visible, observe, radii, grad_accum, grad_accum_abs, denom, max_radii);
    return launched();
}

}  // extern "C"
