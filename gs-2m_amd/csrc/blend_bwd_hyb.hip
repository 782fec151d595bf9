// Backward of the alpha compositing for gfx950, hybrid variant (gs2m_set_bwd_impl(2)).
//
// Semantics: renderCUDA (bwd), diff-gaussian-rasterization/cuda_rasterizer/backward.cu:413-598;
// same tile lists, tests and gradients as blend_bwd.hip (see its header for the scalar-suffix-sum
// form of the per-pixel recurrence and for the no-atomics row scheme).
//
// blend_bwd.hip is VALU-issue bound and about half of its vector instructions per surviving
// (instance, quadrant) pair are the products w*g_ch plus their cross-lane sums.  This variant keeps
// the pixel-per-lane evaluation (including the cheap "no lane contributes -> skip" path) but moves
// those sums to the matrix pipe, which runs beside the VALU:
//   * per contributing survivor every lane stores just two scalars, w = alpha*T and
//     s = opacity*dL/dalpha*G, into a per-wave LDS matrix [survivor][pixel];
//   * every 8 survivors the wave runs 16 k-steps of v_mfma_f32_16x16x4_f32 (exact fp32):
//       W[surv x pixel] * Ggrad[pixel x channel]   -> dL/dcolour, dL/dfeature
//       S[surv x pixel] * Phi[pixel x 6 moments]    -> sum s, s*cx, s*cy, s*cx^2, s*cx*cy, s*cy^2
//     (quadrant-centred pixel coordinates), from which sum(s*dx), sum(s*dx^2), ... follow per survivor;
//   * only the two |.| sums are still reduced across lanes (permlane swaps + 4 DPP adds for both).
// Rows are combined across the four waves in LDS and stored once per tile instance, exactly as in
// blend_bwd.hip, so gaussian_bwd.hip reads the same layout (bitwise reproducible).
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BB = 32;   // instances staged per batch
constexpr int GS = 8;    // survivors per matrix-core group
constexpr int BUFW = 65; // padded row length of the [survivor][pixel] matrices

template <int FC>  // feature channels blended (compile time); runtime fc <= FC
__global__ void __launch_bounds__(256) blend_bwd_hyb_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, int W,
    int H, int tiles_x, const float* __restrict__ bg, int fc, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ grad_color,
    const float* __restrict__ grad_buffer, float* __restrict__ rows, uint8_t* __restrict__ row_valid) {
    constexpr int FQ = (FC + 3) / 4;
    constexpr int NQ = 4 + FQ;
    constexpr int NV = ROW_FEAT + FC;
    constexpr int RQ = (NV + 3) / 4;
    constexpr int ROWF = RQ * 4;
    constexpr int NC = 3 + FC;
    __shared__ float4 s_v[NQ][BB];
    __shared__ uint32_t s_gid[BB];
    __shared__ uint32_t s_slot[BB];
    __shared__ __align__(16) float s_acc[4][BB][ROWF];
    __shared__ float s_bw[4][GS][BUFW];  // w = alpha*T      [wave][survivor][pixel]
    __shared__ float s_bs[4][GS][BUFW];  // s = o*dL/dalpha*G
    __shared__ __align__(16) float s_d[4][GS][8];
    __shared__ int s_gj[4][GS];
    __shared__ unsigned long long s_mask[4];
    __shared__ uint32_t s_max;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.x;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int qx0 = tile_x * GS2M_TILE + (wave & 1) * 8, qy0 = tile_y * GS2M_TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)qx0, bx1 = bx0 + 7.0f, by0 = (float)qy0, by1 = by0 + 7.0f;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;

    const uint2 range = ranges[tile];
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float T = T_final;
    float gpx[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) gpx[k] = 0.f;
    if (inside) {
        gpx[0] = grad_color[pix];
        gpx[1] = grad_color[HW + pix];
        gpx[2] = grad_color[2 * HW + pix];
#pragma unroll
        for (int ch = 0; ch < FC; ch++) gpx[3 + ch] = ch < fc ? grad_buffer[ch * HW + pix] : 0.f;
    }
    // suffix sum seeded with the background term (backward.cu:562-566)
    float Sg = T_final * (bg[0] * gpx[0] + bg[1] * gpx[1] + bg[2] * gpx[2]);
    const float halfW = 0.5f * W, halfH = 0.5f * H;

    // ---- constant B operands of the matrix products: lane (n = lane & 15, k = lane >> 4), k-step t
    // covers pixels 4t..4t+3 of the quadrant (pixel p = lane index of the pixel-per-lane layout) ----
    const int nn = lane & 15, kk = lane >> 4;
    float Bg[16], Bphi[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int p = 4 * t + kk;
        const int bx = qx0 + (p & 7), by = qy0 + (p >> 3);
        float g = 0.f;
        if (bx < W && by < H) {
            const size_t bp = (size_t)by * W + bx;
            if (nn < 3) g = grad_color[(size_t)nn * HW + bp];
            else if (nn - 3 < FC && nn - 3 < fc) g = grad_buffer[(size_t)(nn - 3) * HW + bp];
        }
        Bg[t] = g;
        const float cx = (float)(p & 7) - 3.5f, cy = (float)(p >> 3) - 3.5f;
        Bphi[t] = nn == 0 ? 1.f : nn == 1 ? cx : nn == 2 ? cy : nn == 3 ? cx * cx : nn == 4 ? cx * cy : nn == 5 ? cy * cy : 0.f;
    }
    const float xq = (float)qx0 + 3.5f, yq = (float)qy0 + 3.5f;

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
    int nf = 0;                        // survivors waiting in the [survivor][pixel] matrices
    unsigned long long wrote = 0ull;

    // 16 k-steps over the quadrant's 64 pixels for the nf (<= 8) buffered survivors, then their rows
    auto flush_group = [&]() {
        v4f acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        const bool on = nn < nf;  // A rows >= nf (and rows 8..15) are zero
        const int ri = nn < GS ? nn : 0;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const float aw = on ? s_bw[wave][ri][4 * t + kk] : 0.f;
            const float as = on ? s_bs[wave][ri][4 * t + kk] : 0.f;
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, Bg[t], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(as, Bphi[t], acc2, 0, 0, 0);
        }
        // lane (n = nn, kk) holds D[4*kk + rr][n]; survivors 0..7 live in kk = 0, 1
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int i = 4 * kk + rr;
            if (kk < 2 && i < nf) {
                const int jj = s_gj[wave][i];
                if (nn < NC) s_acc[wave][jj][ROW_COL + nn] = acc1[rr];
                else if (nn < ROWF - ROW_COL) s_acc[wave][jj][ROW_COL + nn] = 0.f;
                if (nn < 6) s_d[wave][i][nn] = acc2[rr];
            }
        }
        // geometry of survivor i from its moments (lanes 0..nf-1); s_d was written by this wave only
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < nf) {
            const int jj = s_gj[wave][lane];
            const float4 a = s_v[REC_GEO0][jj], c = s_v[REC_GEO1][jj];  // x, y, A, B | C, opacity, hx, hy
            const float4 m0 = *reinterpret_cast<const float4*>(&s_d[wave][lane][0]);  // M0, Mx, My, Mxx
            const float2 m1 = *reinterpret_cast<const float2*>(&s_d[wave][lane][4]);  // Mxy, Myy
            const float xc = a.x - xq, yc = a.y - yq;  // dx = xc - cx, dy = yc - cy
            const float Sdx = xc * m0.x - m0.y, Sdy = yc * m0.x - m0.z;
            const float Sdxx = xc * xc * m0.x - 2.f * xc * m0.y + m0.w;
            const float Sdxy = xc * yc * m0.x - xc * m0.z - yc * m0.y + m1.x;
            const float Sdyy = yc * yc * m0.x - 2.f * yc * m0.z + m1.y;
            float* o = &s_acc[wave][jj][0];
            o[0] = -halfW * (a.z * Sdx + a.w * Sdy);
            o[1] = -halfH * (c.x * Sdy + a.w * Sdx);
            o[4] = -0.5f * Sdxx;
            o[5] = -0.5f * Sdxy;
            o[6] = -0.5f * Sdyy;
            o[7] = m0.x != 0.f ? m0.x / c.y : 0.f;
        }
        nf = 0;
    };

    for (int b = nb - 1; b >= -1; b--) {
        gs2m_sync();  // (S1) previous batch fully accumulated
        if (prev_cnt > 0) {
            unsigned long long any = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
            const int q = tid & 7;
            const int row = tid >> 3;  // BB = 32 rows, one pass
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
        if (b < 0) break;
        const int base = b * BB;
        const int cnt = min(BB, maxc - base);
        if (tid < cnt) s_gid[tid] = point_list[range.x + base + tid];
        gs2m_sync();  // (S2)
        {
            const int q = tid & 7, row = tid >> 3;
            if (q < NQ && row < cnt) {
                const float4 v = rec[(size_t)s_gid[row] * REC_Q + q];
                if (q == REC_BIN) {
                    const uint32_t off = f2u(v.x), rm = f2u(v.y), rw = f2u(v.z) & 0xFFFFu;
                    s_slot[row] = off + ((uint32_t)tile_y - (rm >> 16)) * rw + ((uint32_t)tile_x - (rm & 0xFFFFu));
                }
                s_v[q][row] = v;
            }
            if (tid < 4) s_mask[tid] = 0ull;
        }
        gs2m_sync();  // (S3)
        prev_cnt = cnt;

        bool hit = false;
        if (lane < cnt) {
            const float4 a = s_v[REC_GEO0][lane], c = s_v[REC_GEO1][lane];
            hit = gs2m_reaches_rect(a.x, a.y, a.z, a.w, c.x, c.z, c.w, s_v[REC_BIN][lane].w, bx0, bx1, by0, by1);
        }
        unsigned long long mask = __ballot(hit);
        wrote = 0ull;
        while (mask) {
            const int jj = 63 - __builtin_clzll(mask);  // back to front
            mask &= ~(1ull << jj);
            const uint32_t pos = (uint32_t)(base + jj + 1);
            const float4 a = s_v[REC_GEO0][jj], c = s_v[REC_GEO1][jj];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = gs2m_power(dx, dy, a.z, a.w, c.x);
            const float G = gs2m_exp(power);
            const float alpha = fminf(0.99f, c.y * G);
            const bool contrib = (pos <= last) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
            if (__ballot(contrib) == 0ull) continue;

            // branch-free: non-contributing lanes run with alpha = 0, G = 0 (see blend_bwd.hip)
            const float am = contrib ? alpha : 0.f;
            const float Gm = contrib ? G : 0.f;
            const float inv1ma = __builtin_amdgcn_rcpf(1.f - am);
            T = T * inv1ma;
            const float w = am * T;
            const float4 col = s_v[REC_RGB][jj];
            float gc = col.x * gpx[0];
            gc = __builtin_fmaf(col.y, gpx[1], gc);
            gc = __builtin_fmaf(col.z, gpx[2], gc);
#pragma unroll
            for (int q = 0; q < FQ; q++) {
                const float4 f = s_v[REC_FEAT + q][jj];
                const float fa[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * q + e < FC) gc = __builtin_fmaf(fa[e], gpx[3 + 4 * q + e], gc);
            }
            const float dL_dalpha = T * gc - Sg * inv1ma;
            Sg = __builtin_fmaf(gc, w, Sg);
            const float s = c.y * dL_dalpha * Gm;
            const float t1 = dx * a.z + dy * a.w, t2 = dy * c.x + dx * a.w;
            float u1 = fabsf(s * t1), u2 = fabsf(s * t2);
            s_bw[wave][nf][lane] = w;
            s_bs[wave][nf][lane] = s;
            {   // |.| sums over the 64 lanes: rows 0/1 end up with sum(u1), rows 2/3 with sum(u2)
                const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(u1), __float_as_uint(u2), false, false);
                float h = __uint_as_float(x[0]) + __uint_as_float(x[1]);
                const auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(h), __float_as_uint(h), false, false);
                h = __uint_as_float(y[0]) + __uint_as_float(y[1]);
                h += dpp_mov0<DPP_QUAD_XOR1>(h);
                h += dpp_mov0<DPP_QUAD_XOR2>(h);
                h += dpp_mov0<DPP_ROW_HALF_MIRROR>(h);
                h += dpp_mov0<DPP_ROW_MIRROR>(h);
                if (lane == 0) s_acc[wave][jj][2] = halfW * h;
                if (lane == 32) s_acc[wave][jj][3] = halfH * h;
            }
            if (lane == 0) s_gj[wave][nf] = jj;
            wrote |= (1ull << jj);
            nf++;
            if (nf == GS) flush_group();
        }
        if (nf > 0) flush_group();  // rows must be complete before the workgroup combines them
        if (lane == 0) s_mask[wave] = wrote;
    }
}

int fc_template(int fc) { return fc <= 1 ? 1 : (fc <= 5 ? 5 : (fc <= 9 ? 9 : 10)); }

}  // namespace

void gs2m_launch_blend_bwd_hyb(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                               const BinningState& b, const ImageState& im, const float* grad_color,
                               const float* grad_buffer, float* rows, uint8_t* row_valid, hipStream_t s) {
    const int tiles = tiles_x * tiles_y;
#define GS2M_BWDH(FC)                                                                                                 \
    blend_bwd_hyb_kernel<FC><<<tiles, 256, 0, s>>>(im.ranges, b.point_list, g.rec, W, H, tiles_x, bg, fc, im.final_T, \
                                                   im.n_contrib, grad_color, grad_buffer, rows, row_valid)
    switch (fc_template(fc)) {
        case 1: GS2M_BWDH(1); break;
        case 5: GS2M_BWDH(5); break;
        case 9: GS2M_BWDH(9); break;
        default: GS2M_BWDH(10); break;
    }
#undef GS2M_BWDH
}
