// Per-tile back-to-front gradient pass (backward of the alpha compositing) for gfx950.
//
// Semantics: renderCUDA, diff-gaussian-rasterization/cuda_rasterizer/backward.cu:413-598.
// The reference issues (11 + fc) float atomicAdds per contributing pixel x Gaussian pair.  On
// MI355X global float atomics run at ~1.3 TB/s of added bytes and an order of magnitude
// slower when 64 lanes hit 64 rows, so that shape cannot be carried over.  This kernel uses
// NO global atomics:
//   * each wave (one 8x8 pixel quadrant) reduces the 11 + fc partial values of an instance
//     over its 64 lanes with a TRANSPOSED reduction: v_permlane32_swap / v_permlane16_swap
//     halve the register count while doubling the lanes each sum covers (N -> N/4 registers,
//     each 16-lane row then owning different values), followed by 4 DPP adds per remaining
//     register -- ~3 instructions per value instead of 6-10 for independent butterflies;
//     only for instances that survive the quadrant rectangle test AND have a contributing lane;
//   * the four waves' sums are combined in LDS in a fixed order and the workgroup stores one
//     row per (tile, Gaussian) instance with plain coalesced stores, addressed by the
//     instance's EMISSION slot, so all rows of one Gaussian are contiguous in HBM;
//   * gaussian_bwd.hip then sums each Gaussian's contiguous rows (a streaming read) while it
//     runs the rest of the per-Gaussian backward.  Gradients are bitwise reproducible.
// The per-pixel recurrence is restated with one scalar suffix sum: with gc = <g, c_i> over
// the 3 colour + fc feature channels, w_i = alpha_i*T_i and Sg_i = sum_{j>i} gc_j*w_j
// (+ T_final*<bg, g_colour>), dL/dalpha_i = T_i*gc_i - Sg_i/(1 - alpha_i), which equals
// backward.cu:541-566 (accum_rec/last_color per channel) term by term.
#include "common.h"

namespace {

constexpr int BB = 64;  // instances per batch: one per lane in the quadrant test

// Sum N (multiple of 4) per-lane values over the 64 lanes of the wave.  On return r[q]
// (q < N/4) holds, in every lane of 16-lane row `row`, the total of value q + (N/4)*row.
template <int N>
__device__ __forceinline__ void wave_reduce_transposed(float (&v)[N], float (&r)[N / 4]) {
    static_assert(N % 4 == 0, "N must be a multiple of 4");
    constexpr int H = N / 2, Q = N / 4;
    float u[H];
#pragma unroll
    for (int i = 0; i < H; i++) {  // lanes 0-31 <- value i, lanes 32-63 <- value i + H (each summed over l, l^32)
        const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + H]), false, false);
        u[i] = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    }
#pragma unroll
    for (int i = 0; i < Q; i++) {  // row0 <- i, row1 <- i + Q, row2 <- i + H, row3 <- i + H + Q
        const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[i]), __float_as_uint(u[i + Q]), false, false);
        float w = __uint_as_float(t[0]) + __uint_as_float(t[1]);
        w += dpp_mov0<DPP_QUAD_XOR1>(w);
        w += dpp_mov0<DPP_QUAD_XOR2>(w);
        w += dpp_mov0<DPP_ROW_HALF_MIRROR>(w);
        w += dpp_mov0<DPP_ROW_MIRROR>(w);
        r[i] = w;
    }
}

template <int FC>  // feature channels blended (compile time); runtime fc <= FC
__global__ void __launch_bounds__(256) blend_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, int W,
    int H, int tiles_x, const float* __restrict__ bg, int fc, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ grad_color,
    const float* __restrict__ grad_buffer, float* __restrict__ rows, uint8_t* __restrict__ row_valid) {
    constexpr int FQ = (FC + 3) / 4;
    constexpr int NQ = 3 + (3 + FC + 3) / 4;  // geo0, geo1, bin + channel quads
    constexpr int NV = ROW_FEAT + FC;  // values reduced per instance
    constexpr int RQ = (NV + 3) / 4;
    constexpr int ROWF = RQ * 4;       // row length in floats (zero padded)
    __shared__ float4 s_v[NQ][BB];
    __shared__ uint32_t s_gid[BB];
    __shared__ uint32_t s_slot[BB];
    __shared__ __align__(16) float s_acc[4][BB][ROWF];
    __shared__ unsigned long long s_mask[4];
    __shared__ uint32_t s_max;
    __shared__ float s_dummy[256];  // sink for the lanes that own no value after the reduction

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.x;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int px = tile_x * GS2M_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tile_y * GS2M_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)(tile_x * GS2M_TILE + (wave & 1) * 8), bx1 = bx0 + 7.0f;
    const float by0 = (float)(tile_y * GS2M_TILE + (wave >> 1) * 8), by1 = by0 + 7.0f;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;

    const uint2 range = ranges[tile];
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float T = T_final;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    float4 gf[FQ];
#pragma unroll
    for (int q = 0; q < FQ; q++) gf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) {
        g0 = grad_color[pix];
        g1 = grad_color[HW + pix];
        g2 = grad_color[2 * HW + pix];
        float t[12];
#pragma unroll
        for (int ch = 0; ch < 12; ch++) t[ch] = (ch < fc && ch < FC) ? grad_buffer[ch * HW + pix] : 0.f;
#pragma unroll
        for (int q = 0; q < FQ; q++) gf[q] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
    }
    // suffix sum seeded with the background term (backward.cu:562-566)
    float Sg = T_final * (bg[0] * g0 + bg[1] * g1 + bg[2] * g2);
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // entries past the tile's largest n_contrib are never touched by any pixel
    if (tid == 0) s_max = 0;
    gs2m_sync();
    {
        uint32_t m = last;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        if (lane == 0) atomicMax(&s_max, m);
    }
    gs2m_sync();
    const int maxc = (int)s_max;
    const int nb = (maxc + BB - 1) / BB;
    int prev_cnt = 0;

    for (int b = nb - 1; b >= -1; b--) {
        gs2m_sync();  // (S1) previous batch fully accumulated
        if (prev_cnt > 0) {
            // write the rows of the previous batch: 8 lanes x 16 B per row, fixed wave order
            unsigned long long any = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
            const int q = tid & 7;
#pragma unroll
            for (int r = 0; r < BB / 32; r++) {
                const int row = r * 32 + (tid >> 3);
                if (row < prev_cnt && ((any >> row) & 1ull)) {
                    const uint32_t slot = s_slot[row];
                    if (q < RQ) {
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            if ((s_mask[w] >> row) & 1ull) {
                                const float4 v = *reinterpret_cast<const float4*>(&s_acc[w][row][4 * q]);
                                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                            }
                        }
                        reinterpret_cast<float4*>(rows + (size_t)slot * ROWF)[q] = acc;
                    }
                    if (q == 7) row_valid[slot] = 1;
                }
            }
        }
        if (b < 0) break;
        const int base = b * BB;
        const int cnt = min(BB, maxc - base);
        if (tid < cnt) s_gid[tid] = point_list[range.x + base + tid];
        gs2m_sync();  // (S2)
        {
            const int q = tid & 7;
            if (q < NQ) {
#pragma unroll
                for (int r = 0; r < BB / 32; r++) {
                    const int row = r * 32 + (tid >> 3);
                    if (row < cnt) {
                        float4 v = rec[(size_t)s_gid[row] * REC_Q + q];
                        if (q == REC_BIN) {
                            const uint32_t off = f2u(v.x), rm = f2u(v.y), rw = f2u(v.z) & 0xFFFFu;
                            s_slot[row] = off + ((uint32_t)tile_y - (rm >> 16)) * rw + ((uint32_t)tile_x - (rm & 0xFFFFu));
                        }
                        s_v[q][row] = v;
                    }
                }
            }
            if (tid < 4) s_mask[tid] = 0ull;
        }
        gs2m_sync();  // (S3)
        prev_cnt = cnt;

        bool hit = false;
        if (lane < cnt) {
            const float4 a = s_v[REC_GEO0][lane], c = s_v[REC_GEO1][lane];
            hit = gs2m_reaches_rect(a.x, a.y, a.z, a.w, c.x, s_v[REC_BIN][lane].w, bx0, bx1, by0, by1);
        }
        unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        unsigned long long wrote = 0ull;
        while (mask) {
            const int jj = 63 - __builtin_clzll(mask);  // back to front
            mask &= ~(1ull << jj);
            const uint32_t pos = (uint32_t)(base + jj + 1);  // 1-based list position
            const float4 a = s_v[REC_GEO0][jj], c = s_v[REC_GEO1][jj];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float p2 = gs2m_power(dx, dy, a.z, a.w, c.x);
            const float G = gs2m_exp(p2);
            const float alpha = fminf(0.99f, c.y * G);
            const bool contrib = (pos <= last) && (p2 <= 0.0f) && (alpha >= 1.0f / 255.0f);
            if (__builtin_amdgcn_ballot_w64(contrib) == 0ull) continue;

            // Branch-free payload: non-contributing lanes run with alpha = 0 and G = 0, which makes every
            // reduced value exactly 0 and leaves T and Sg unchanged (rcp(1) = 1, fma(gc, 0, Sg) = Sg).
            float v[ROWF];
#pragma unroll
            for (int k = NV; k < ROWF; k++) v[k] = 0.f;
            {
                const float am = contrib ? alpha : 0.f;
                const float Gm = contrib ? G : 0.f;
                const float inv1ma = __builtin_amdgcn_rcpf(1.f - am);
                T = T * inv1ma;
                const float w = am * T;
                float chv[4 * (NQ - 3)];  // r, g, b, feature 0.. (record quads REC_CH..)
#pragma unroll
                for (int q = 0; q < NQ - 3; q++) {
                    const float4 t = s_v[REC_CH + q][jj];
                    chv[4 * q] = t.x; chv[4 * q + 1] = t.y; chv[4 * q + 2] = t.z; chv[4 * q + 3] = t.w;
                }
                const float4 col = make_float4(chv[0], chv[1], chv[2], 0.f);
                float gc = col.x * g0;
                gc = __builtin_fmaf(col.y, g1, gc);
                gc = __builtin_fmaf(col.z, g2, gc);
                v[ROW_COL + 0] = w * g0;
                v[ROW_COL + 1] = w * g1;
                v[ROW_COL + 2] = w * g2;
#pragma unroll
                for (int q = 0; q < FQ; q++) {
                    float fa[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) fa[e] = 4 * q + e < FC ? chv[3 + 4 * q + e] : 0.f;
                    const float ga[4] = {gf[q].x, gf[q].y, gf[q].z, gf[q].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (4 * q + e < FC) {
                            gc = __builtin_fmaf(fa[e], ga[e], gc);
                            v[ROW_FEAT + 4 * q + e] = w * ga[e];
                        }
                    }
                }
                const float dL_dalpha = T * gc - Sg * inv1ma;
                Sg = __builtin_fmaf(gc, w, Sg);
                // a = x, y, A, B;  c = C, opacity, hx, hy.  With s = opacity * dL/dalpha * G (backward.cu:569-595):
                //   d mean2D = -s * (A dx + B dy, C dy + B dx) * (W/2, H/2),  d conic = -0.5 s (dx^2, dx dy, dy^2)
                const float GdA = Gm * dL_dalpha;
                const float s_ = c.y * GdA;
                const float t1 = __builtin_fmaf(a.w, dy, a.z * dx), t2 = __builtin_fmaf(a.w, dx, c.x * dy);
                const float mx = (s_ * t1) * -ddelx_dx;
                const float my = (s_ * t2) * -ddely_dy;
                const float hs = -0.5f * s_;
                const float hsdx = hs * dx;
                v[0] = mx;
                v[1] = my;
                v[2] = fabsf(mx);
                v[3] = fabsf(my);
                v[4] = hsdx * dx;
                v[5] = hsdx * dy;
                v[6] = (hs * dy) * dy;
                v[7] = GdA;
            }
            // transposed 64-lane sums: lane (row, c) with c < RQ ends up owning value c + RQ*row
            float r[RQ];
            wave_reduce_transposed<ROWF>(v, r);
            const int c16 = lane & 15, row = lane >> 4;
            float outv = r[0];
#pragma unroll
            for (int q = 1; q < RQ; q++) outv = (c16 == q) ? r[q] : outv;
            // every lane stores (owners into the row, the others into a sink): keeping the store out of a
            // divergent branch lets the last DPP add of the reduction stay a single v_add_f32_dpp
            float* dst = (c16 < RQ) ? &s_acc[wave][jj][c16 + RQ * row] : &s_dummy[tid];
            *dst = outv;
            wrote |= (1ull << jj);
        }
        if (lane == 0) s_mask[wave] = wrote;
    }
}

}  // namespace

static int fc_template(int fc) { return fc <= 1 ? 1 : (fc <= 5 ? 5 : (fc <= 9 ? 9 : 10)); }

int gs2m_row_floats(int fc) { return ((ROW_FEAT + fc_template(fc) + 3) / 4) * 4; }

void gs2m_launch_blend_bwd(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                           const BinningState& b, const ImageState& im, const float* grad_color,
                           const float* grad_buffer, float* rows, uint8_t* row_valid, hipStream_t s) {
    const int tiles = tiles_x * tiles_y;
#define GS2M_BWD(FC)                                                                                              \
    blend_bwd_kernel<FC><<<tiles, 256, 0, s>>>(im.ranges, b.point_list, g.rec, W, H, tiles_x, bg, fc, im.final_T, \
                                               im.n_contrib, grad_color, grad_buffer, rows, row_valid)
    switch (fc_template(fc)) {
        case 1: GS2M_BWD(1); break;
        case 5: GS2M_BWD(5); break;
        case 9: GS2M_BWD(9); break;
        default: GS2M_BWD(10); break;
    }
#undef GS2M_BWD
}
