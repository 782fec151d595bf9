// Per-Gaussian forward preprocessing for gfx950: cull, project, 3D->2D covariance, conic,
// radius, tile rectangle, SH->RGB, and the 128-byte blend record the tile kernels gather.
//
// Behaviour follows the reference kernel preprocessCUDA
// (diff-gaussian-rasterization/cuda_rasterizer/forward.cu:145-241 with computeCov3D :109-142,
// computeCov2D :70-104, computeColorFromSH :20-67, in_frustum auxiliary.h:140-162,
// ndc2Pix/getRect auxiliary.h:40-53) but the data layout is different: instead of six SoA
// arrays it emits ONE aligned 128-B record per visible Gaussian (common.h) plus the
// fp32-bit depth key the tiles' lists are ordered by.  This file is compiled with
// -ffp-contract=off: the values that decide integers (radius, tile rect, depth key) are
// evaluated in exactly the written order.
#include "common.h"

namespace {

struct M3 {  // column-major 3x3, m[col][row]
    float m[3][3];
};
__device__ __forceinline__ M3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    M3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ M3 m3_t(const M3& A) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
    return R;
}

__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(hi, max(lo, v)); }

template <bool SH_LDS>
__global__ void __launch_bounds__(256) preprocess_kernel(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ shs_rest, const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const float* __restrict__ features, const float* __restrict__ vm, const float* __restrict__ pm,
    const float* __restrict__ cam_pos, int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y,
    int tiles_x, int tiles_y, int shrink, int* __restrict__ radii, int* __restrict__ observe_zero, float4* __restrict__ rec,
    uint32_t* __restrict__ tiles_touched, uint2* __restrict__ rect, uint32_t* __restrict__ block_tt, uint32_t* __restrict__ block_hu,
    uint32_t* __restrict__ depth_key, uint8_t* __restrict__ clamped, float* __restrict__ sh_dir, ZeroJobs zero) {
    // SH rows go through LDS (common.h: gs2m_stage_sh); other M fall back to direct per-thread loads.
    // (the block's blend records are parked in the same LDS afterwards: 256 x 36 floats)
    __shared__ __align__(16) float s_sh[SH_LDS ? 256 * 49 : 256 * 45];  // (afterwards: 256 x 36 floats of records + 256 x 9 of sh_dir)
    __shared__ uint8_t s_seen[256];  // the thread's Gaussian has a radius: its record is stored
    __shared__ uint32_t s_tt[4];     // tiles_touched summed per wave: the block's total is the binning's block sum (binning.hip)
    __shared__ uint32_t s_hu[4];     // heavy units per wave (common.h: GS2M_HEAVY_TILES)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // The thread's own inputs are requested BEFORE the block stages its SH rows: the staging ends in a barrier, and loads
    // issued behind it would cost a second exposed memory round trip (the kernel is latency bound at 12 waves per CU).
    const bool inr = idx < P;
    const int li = inr ? idx : 0;
    const float px = means3D[3 * li], py = means3D[3 * li + 1], pz = means3D[3 * li + 2];
    float in_s[3] = {0.f, 0.f, 0.f};
    float4 in_q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cov3D_precomp == nullptr) {
        in_s[0] = scales[3 * li]; in_s[1] = scales[3 * li + 1]; in_s[2] = scales[3 * li + 2];
        in_q = reinterpret_cast<const float4*>(rotations)[li];
    }
    const float in_op = opacities[li];
    float2 in_f[GS2M_NUM_FEATURES / 2];
#pragma unroll
    for (int k = 0; k < GS2M_NUM_FEATURES / 2; k++)
        in_f[k] = features != nullptr ? reinterpret_cast<const float2*>(features + (size_t)li * GS2M_NUM_FEATURES)[k] : make_float2(0.f, 0.f);
    if (SH_LDS) {
        gs2m_stage_sh(shs, shs_rest, P, s_sh);
        gs2m_sync();
    }
    gs2m_zero_jobs(zero, (size_t)idx, (size_t)gridDim.x * blockDim.x);  // the sort / scan scratch of the stages that follow
    int out_radius = 0;
    float sd[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // d(SH colour)/d(direction): the backward's only use of the coefficients
    uint32_t out_tt = 0;
    uint32_t out_key = 0xFFFFFFFFu;
    uint2 out_rect = make_uint2(0u, 0u);

    // view-space point (transformPoint4x3), near-plane cull at 0.2
    const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    float4 rq[REC_Q];  // the blend record (zeros for a Gaussian that emits nothing: its record is never read)
#pragma unroll
    for (int k = 0; k < REC_Q; k++) rq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inr && vz > 0.2f) {
        const float hx_ = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
        const float hy_ = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
        const float hw_ = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
        const float p_w = 1.0f / (hw_ + 0.0000001f);
        const float projx = hx_ * p_w, projy = hy_ * p_w;

        float c3[6];
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * (size_t)idx + k];
        } else {
            const float sx = scale_modifier * in_s[0], sy = scale_modifier * in_s[1], sz = scale_modifier * in_s[2];
            const float4 q = in_q;
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            M3 S = m3_cols(sx, 0.f, 0.f, 0.f, sy, 0.f, 0.f, 0.f, sz);
            M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                           2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                           2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            M3 Mm = m3_mul(S, R);
            M3 Sig = m3_mul(m3_t(Mm), Mm);
            c3[0] = Sig.m[0][0]; c3[1] = Sig.m[0][1]; c3[2] = Sig.m[0][2];
            c3[3] = Sig.m[1][1]; c3[4] = Sig.m[1][2]; c3[5] = Sig.m[2][2];
        }
        // EWA projection of the covariance (no +0.3 low-pass in this fork's forward)
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        M3 J = m3_cols(focal_x / vz, 0.0f, -(focal_x * tx) / (vz * vz), 0.0f, focal_y / vz, -(focal_y * ty) / (vz * vz),
                       0.f, 0.f, 0.f);
        M3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
        M3 T = m3_mul(Wm, J);
        M3 Vrk = m3_cols(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
        M3 cov = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
        const float cova = cov.m[0][0], covb = cov.m[0][1], covc = cov.m[1][1];
        const float det = cova * covc - covb * covb;
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float cA = covc * det_inv, cB = -covb * det_inv, cC = cova * det_inv;
            const float mid = 0.5f * (cova + covc);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix is evaluated in double in the reference (auxiliary.h:40-42)
            const float pix = (float)((((double)projx + 1.0) * (double)W - 1.0) * 0.5);
            const float piy = (float)((((double)projy + 1.0) * (double)H - 1.0) * 0.5);
            const int mr = (int)radius;
            const int rminx = clampi((int)((pix - mr) / GS2M_TILE), 0, tiles_x);
            const int rminy = clampi((int)((piy - mr) / GS2M_TILE), 0, tiles_y);
            const int rmaxx = clampi((int)((pix + mr + GS2M_TILE - 1) / GS2M_TILE), 0, tiles_x);
            const int rmaxy = clampi((int)((piy + mr + GS2M_TILE - 1) / GS2M_TILE), 0, tiles_y);
            const uint32_t rw = (uint32_t)(rmaxx - rminx), rh = (uint32_t)(rmaxy - rminy);
            if (rw * rh != 0) {
                float cr, cg, cb;
                uint8_t cl = 0;
                if (colors_precomp == nullptr) {
                    float dx = px - cam_pos[0], dy = py - cam_pos[1], dz = pz - cam_pos[2];
                    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                    dx = dx / len; dy = dy / len; dz = dz / len;
                    const float x = dx, y = dy, z = dz;
                    const float* sh = SH_LDS ? (s_sh + threadIdx.x * 49) : (shs + (size_t)idx * M * 3);
                    float res[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        float r_ = SH_C0 * sh[c];
                        if (D > 0) {
                            r_ = r_ - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
                            if (D > 1) {
                                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                                r_ = r_ + kSH_C2[0] * xy * sh[12 + c] + kSH_C2[1] * yz * sh[15 + c] +
                                     kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] + kSH_C2[3] * xz * sh[21 + c] +
                                     kSH_C2[4] * (xx - yy) * sh[24 + c];
                                if (D > 2) {
                                    r_ = r_ + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] +
                                         kSH_C3[1] * xy * z * sh[30 + c] +
                                         kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                                         kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                                         kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] +
                                         kSH_C3[5] * z * (xx - yy) * sh[42 + c] +
                                         kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                                }
                            }
                        }
                        r_ += 0.5f;
                        if (r_ < 0) cl |= (uint8_t)(1u << c);
                        res[c] = fmaxf(r_, 0.0f);
                    }
                    cr = res[0]; cg = res[1]; cb = res[2];
                    // The derivative of the colour with respect to the view direction (backward.cu:62-146), taken here where
                    // the 48 coefficients are at hand: the per-Gaussian backward then reads these 36 bytes instead of the
                    // 192-byte SH row.  Same expressions in the same order as the reference's backward: the bits it would get.
                    if (D > 0) {
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            float ddx_ = -SH_C1 * sh[9 + c], ddy_ = -SH_C1 * sh[3 + c], ddz_ = SH_C1 * sh[6 + c];
                            if (D > 1) {
                                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                                ddx_ += kSH_C2[0] * y * sh[12 + c] + kSH_C2[2] * 2.f * -x * sh[18 + c] + kSH_C2[3] * z * sh[21 + c] + kSH_C2[4] * 2.f * x * sh[24 + c];
                                ddy_ += kSH_C2[0] * x * sh[12 + c] + kSH_C2[1] * z * sh[15 + c] + kSH_C2[2] * 2.f * -y * sh[18 + c] + kSH_C2[4] * 2.f * -y * sh[24 + c];
                                ddz_ += kSH_C2[1] * y * sh[15 + c] + kSH_C2[2] * 2.f * 2.f * z * sh[18 + c] + kSH_C2[3] * x * sh[21 + c];
                                if (D > 2) {
                                    ddx_ += (kSH_C3[0] * sh[27 + c] * 3.f * 2.f * xy + kSH_C3[1] * sh[30 + c] * yz +
                                             kSH_C3[2] * sh[33 + c] * -2.f * xy + kSH_C3[3] * sh[36 + c] * -3.f * 2.f * xz +
                                             kSH_C3[4] * sh[39 + c] * (-3.f * xx + 4.f * zz - yy) +
                                             kSH_C3[5] * sh[42 + c] * 2.f * xz + kSH_C3[6] * sh[45 + c] * 3.f * (xx - yy));
                                    ddy_ += (kSH_C3[0] * sh[27 + c] * 3.f * (xx - yy) + kSH_C3[1] * sh[30 + c] * xz +
                                             kSH_C3[2] * sh[33 + c] * (-3.f * yy + 4.f * zz - xx) +
                                             kSH_C3[3] * sh[36 + c] * -3.f * 2.f * yz + kSH_C3[4] * sh[39 + c] * -2.f * xy +
                                             kSH_C3[5] * sh[42 + c] * -2.f * yz + kSH_C3[6] * sh[45 + c] * -3.f * 2.f * xy);
                                    ddz_ += (kSH_C3[1] * sh[30 + c] * xy + kSH_C3[2] * sh[33 + c] * 4.f * 2.f * yz +
                                             kSH_C3[3] * sh[36 + c] * 3.f * (2.f * zz - xx - yy) +
                                             kSH_C3[4] * sh[39 + c] * 4.f * 2.f * xz + kSH_C3[5] * sh[42 + c] * (xx - yy));
                                }
                            }
                            sd[c] = ddx_; sd[3 + c] = ddy_; sd[6 + c] = ddz_;
                        }
                    }
                } else {
                    cr = colors_precomp[3 * idx]; cg = colors_precomp[3 * idx + 1]; cb = colors_precomp[3 * idx + 2];
                }
                clamped[idx] = cl;

                // Half extents of the region where alpha = opacity*exp(power) can reach 1/255,
                // used by the blend kernels to skip 8x8 pixel blocks.  Must never under-estimate:
                // evaluated in double from the very conic the blend kernels use, inflated, and
                // switched off (infinite) for ill-conditioned or indefinite conics.
                const float op = in_op;
                float ex, ey;
                float tau2f = __builtin_inff();  // bound on A dx^2 + 2B dx dy + C dy^2 inside the ellipse
                if (op < 1.0f / 255.0f) {
                    ex = ey = -1.0f;  // alpha <= opacity < 1/255 everywhere: never contributes
                    tau2f = -1.0f;
                } else {
                    const double dA = cA, dB = cB, dC = cC;
                    const double ddet = dA * dC - dB * dB;
                    if (!(ddet > 0.0) || !(dA > 0.0) || !(dC > 0.0) || dA * dC > 1.0e4 * ddet) {
                        ex = ey = __builtin_inff();
                    } else {
                        const double tau2 = 2.0 * fmax(0.0, log(255.0 * (double)op) + 1.0e-3);
                        ex = (float)(sqrt(tau2 * dC / ddet) * 1.001 + 0.01);
                        ey = (float)(sqrt(tau2 * dA / ddet) * 1.001 + 0.01);
                        tau2f = (float)(tau2 * 1.002 + 1.0e-3);
                    }
                }
                // Tile rectangle actually emitted: the reference's radius rectangle intersected with
                // the tiles the alpha >= 1/255 ellipse's bounding box can reach (pixel centres of tile t
                // are 16t .. 16t+15).  Dropped tiles hold no contributing pixel, so every output is
                // unchanged while ~30 % fewer instances are sorted, staged and reduced.  `radii` keeps
                // the reference value.  shrink = 0 reproduces the reference's lists exactly.
                int ex0 = rminx, ex1 = rmaxx, ey0 = rminy, ey1 = rmaxy;
                if (shrink) {
                    if (ex < 0.f) {
                        ex1 = ex0; ey1 = ey0;
                    } else {
                        const float lx = ceilf((pix - ex - 15.0f) * 0.0625f), hx2 = floorf((pix + ex) * 0.0625f) + 1.0f;
                        const float ly = ceilf((piy - ey - 15.0f) * 0.0625f), hy2 = floorf((piy + ey) * 0.0625f) + 1.0f;
                        ex0 = max(ex0, (int)fminf(fmaxf(lx, -1.0f), 70000.0f));
                        ex1 = min(ex1, (int)fminf(fmaxf(hx2, -1.0f), 70000.0f));
                        ey0 = max(ey0, (int)fminf(fmaxf(ly, -1.0f), 70000.0f));
                        ey1 = min(ey1, (int)fminf(fmaxf(hy2, -1.0f), 70000.0f));
                        if (ex1 < ex0) ex1 = ex0;
                        if (ey1 < ey0) ey1 = ey0;
                    }
                }
                const uint32_t ew = (uint32_t)(ex1 - ex0), eh = (uint32_t)(ey1 - ey0);
                float4* r4 = rq;
                r4[REC_GEO0] = make_float4(pix, piy, cA, cB);
                r4[REC_GEO1] = make_float4(cC, op, ex, ey);
                r4[REC_BIN] = make_float4(u2f(0u), u2f((uint32_t)ex0 | ((uint32_t)ey0 << 16)), u2f(ew | (eh << 16)), tau2f);
                float f[GS2M_NUM_FEATURES];
#pragma unroll
                for (int k = 0; k < GS2M_NUM_FEATURES; k++) f[k] = 0.f;
#pragma unroll
                for (int k = 0; k < GS2M_NUM_FEATURES / 2; k++) {
                    f[2 * k] = in_f[k].x; f[2 * k + 1] = in_f[k].y;
                }
                r4[REC_CH + 0] = make_float4(cr, cg, cb, f[0]);
                r4[REC_CH + 1] = make_float4(f[1], f[2], f[3], f[4]);
                r4[REC_CH + 2] = make_float4(f[5], f[6], f[7], f[8]);
                r4[REC_CH + 3] = make_float4(f[9], 0.f, 0.f, 0.f);
                out_radius = mr;
                out_tt = ew * eh;  // 0 is possible: visible (radii > 0) but nothing to emit
                if (out_tt != 0u) out_rect = make_uint2((uint32_t)ex0 | ((uint32_t)ey0 << 16), ew | (eh << 16));
                out_key = f2u(vz);
            }
        }
    }
    if (inr) {
        radii[idx] = out_radius;
        if (observe_zero) observe_zero[idx] = 0;  // the forward adds its counts with integer atomics
        tiles_touched[idx] = out_tt;
        rect[idx] = out_rect;
        depth_key[idx] = out_key;
    }
    {   // the block's instance count: what the binning adds up in front of a block instead of running a scan over P
        const uint32_t ws = wave_inclusive_scan_u32(out_tt, (int)(threadIdx.x & 63));
        if ((threadIdx.x & 63) == 63) s_tt[threadIdx.x >> 6] = ws;
        uint32_t hu = gs2m_heavy(out_tt, GS2M_CROWDED_WAVE) ? (out_tt + GS2M_UNIT - 1u) / GS2M_UNIT : 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) hu += __shfl_xor(hu, d, 64);
        if ((threadIdx.x & 63) == 0) s_hu[threadIdx.x >> 6] = hu;
    }
    // The block's 256 records are one contiguous 32-KB run of `rec`: they leave through LDS (row stride 36 floats:
    // conflict-free 16-B writes) as fully coalesced float4 stores -- a thread storing its own record writes seven 16-B
    // pieces into a line of its own, 64 lines per store instruction.  Records of Gaussians without a radius (culled,
    // behind the near plane, degenerate) are not stored: nothing downstream reads them (emit_kernel and the blend lists
    // only reach Gaussians with tiles) -- 128 bytes per unseen Gaussian, most of the scene in a typical training view.
    gs2m_sync();  // every thread is done with its SH row
    float4* s_rec = reinterpret_cast<float4*>(s_sh);
#pragma unroll
    for (int k = 0; k < REC_Q; k++) s_rec[threadIdx.x * 9 + k] = rq[k];
    s_seen[threadIdx.x] = out_radius > 0 ? 1 : 0;
    if (threadIdx.x == 0) {  // (a barrier lies between the writes and these reads)
        block_tt[blockIdx.x] = s_tt[0] + s_tt[1] + s_tt[2] + s_tt[3];
        block_hu[blockIdx.x] = s_hu[0] + s_hu[1] + s_hu[2] + s_hu[3];
    }
    float* s_sd = s_sh + 256 * 36;  // the block's 256 x 9 direction derivatives: one contiguous 9-KB run of `sh_dir`
    if (colors_precomp == nullptr) {
#pragma unroll
        for (int k = 0; k < 9; k++) s_sd[threadIdx.x * 9 + k] = sd[k];
    }
    gs2m_sync();
    const size_t lim4 = (size_t)P * REC_Q, base4 = (size_t)blockIdx.x * 256 * REC_Q;
#pragma unroll
    for (int k = 0; k < REC_Q; k++) {
        const int e = k * 256 + threadIdx.x;
        if (base4 + e < lim4 && s_seen[e >> 3]) rec[base4 + e] = s_rec[(e >> 3) * 9 + (e & 7)];
    }
    if (colors_precomp == nullptr) {  // as whole float4s (zeros for Gaussians without a radius), the array's last few floats one by one
        const size_t limf = (size_t)P * 9, basef = (size_t)blockIdx.x * 256 * 9;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e4 = k * 256 + threadIdx.x;  // float4 number inside the block's run (576 of them)
            if (e4 < 576) {
                const size_t f0 = basef + 4 * (size_t)e4;
                if (f0 + 3 < limf) {
                    reinterpret_cast<float4*>(sh_dir + basef)[e4] = reinterpret_cast<const float4*>(s_sd)[e4];
                } else {
                    for (int t = 0; t < 4; t++)
                        if (f0 + t < limf) sh_dir[f0 + t] = s_sd[4 * e4 + t];
                }
            }
        }
    }
}

// markVisible / checkFrustum (rasterizer_impl.cu:48-59, 132-143)
__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm,
                                    uint8_t* __restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    present[idx] = vz <= 0.2f ? 0 : 1;
}

// `prefiltered` (auxiliary.h:155-158): the caller promises that no Gaussian is behind the near plane; the reference traps
// when one is.  Launched only when the flag is set: any such Gaussian raises `flag` (a mapped pinned word the host reads
// after its num_rendered wait).
__global__ void prefiltered_check_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm, uint32_t* flag) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (idx < P) {
        const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
        bad = !(vm[2] * px + vm[6] * py + vm[10] * pz + vm[14] > 0.2f);  // the complement of preprocess_kernel's test
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0)
        __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

void gs2m_launch_prefiltered_check(int P, const float* means3D, const float* viewmatrix, uint32_t* flag, hipStream_t s) {
    prefiltered_check_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, means3D, viewmatrix, flag);
}

void gs2m_launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                            const float* rotations, const float* opacities, const float* shs, const float* shs_rest,
                            const float* cov3D_precomp, const float* colors_precomp, const float* features,
                            const float* viewmatrix, const float* projmatrix, const float* cam_pos, int W, int H,
                            float tan_fovx, float tan_fovy, float focal_x, float focal_y, int tiles_x, int tiles_y,
                            int* radii, int* observe_zero, const GeomState& g, int shrink, const ZeroJobs& zero, hipStream_t s) {
#define GS2M_PRE(LDS)                                                                                                  \
    preprocess_kernel<LDS><<<(P + 255) / 256, 256, 0, s>>>(P, D, M, means3D, scales, scale_modifier, rotations, opacities, \
                                                           shs, shs_rest, cov3D_precomp, colors_precomp, features, viewmatrix,      \
                                                           projmatrix, cam_pos, W, H, tan_fovx, tan_fovy, focal_x,        \
                                                           focal_y, tiles_x, tiles_y, shrink, radii, observe_zero, g.rec,             \
                                                           g.tiles_touched, g.rect, g.block_tt, g.block_hu, g.depth_key, g.clamped, g.sh_dir, zero)
    // split SH (shs = DC, shs_rest = the other 15 coefficients) exists in the LDS-staged form only: api.hip checks
    const bool lds = colors_precomp == nullptr && shs != nullptr && M == 16 &&
                     (shs_rest ? (((uintptr_t)shs_rest) & 15) == 0 : (((uintptr_t)shs) & 15) == 0);
    if (lds) GS2M_PRE(true);
    else GS2M_PRE(false);
#undef GS2M_PRE
}

void gs2m_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s) {
    mark_visible_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, means3D, viewmatrix, present);
}
