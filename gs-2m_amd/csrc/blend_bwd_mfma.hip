// Backward of the alpha compositing for gfx950, matrix-core variant.
//
// Semantics: renderCUDA (bwd), diff-gaussian-rasterization/cuda_rasterizer/backward.cu:413-598 --
// same tile lists, same per-pixel tests, same gradients as blend_bwd.hip, which remains the
// reference implementation inside this library (gs2m_set_bwd_impl).
//
// Why another kernel: rocprofv3 shows the pixel-per-lane backward is VALU-issue bound, and a third
// of its instructions are the cross-lane reduction of the 11+fc per-Gaussian sums.  Here the lanes
// are turned around:
//   lane l = (survivor j = l & 15, pixel row r = l >> 4);  step t = 0..15 covers pixels 4t..4t+3
// i.e. a wave evaluates 16 surviving Gaussians x 4 pixels per instruction and walks the 64 pixels
// of its 8x8 quadrant in 16 steps.  Then
//   * the per-pixel recurrences (transmittance T_i = T/(1-alpha_i), suffix sum Sg_i) run ACROSS
//     the 16 survivor lanes of a DPP row: Kogge-Stone scans with row_shr:1/2/4/8 (product and
//     sum), row totals broadcast with ds_swizzle;
//   * the per-Gaussian sums over pixels are exact-fp32 matrix products on the MFMA pipe
//     (v_mfma_f32_16x16x4_f32, A = per-lane value laid out [survivor][pixel], B = per-pixel
//     constants):  dL/dcolour,feature = W x Ggrad;  geometry = S x Phi with
//     Phi = [1, cx, cy, cx^2, cx*cy, cy^2] (quadrant-centred pixel coordinates), from which
//     sum(s*dx), sum(s*dx^2), ... follow per Gaussian;  |.| sums = U x e_k.
//     The matrix pipe runs beside the VALU, so the reductions cost no vector issue slots.
// One wave per quadrant (64-thread workgroups, no cross-wave barriers), XCD-aware block ids, one
// partial-gradient row per (instance, quadrant) -- gaussian_bwd.hip sums the 4 quadrant rows.
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BB = 32;  // instances staged per batch (LDS per wave decides the occupancy here)
#ifndef GS2M_BWDM_UNROLL_B
#define GS2M_BWDM_UNROLL_B 1
#endif

// Inclusive prefix product over the 16 lanes of a row.  One v_mul_f32_dpp per level: lanes whose source
// falls outside the row are disabled by the DPP (bound_ctrl:0) and keep x, i.e. multiply by 1.  hipcc does
// not fold mov_dpp + mul for a float identity, hence the asm; `s_nop 1` covers the VALU-write -> DPP-read
// hazard, which the compiler does not track through inline asm.
__device__ __forceinline__ float row_scan_mul(float x) {
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}
__device__ __forceinline__ float row_scan_add(float x) {  // inclusive prefix sum over the 16 lanes of a row
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_ROW_SHR(1), 0xF, 0xF, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_ROW_SHR(2), 0xF, 0xF, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_ROW_SHR(4), 0xF, 0xF, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_ROW_SHR(8), 0xF, 0xF, false));
    return x;
}

template <int FC>
__global__ void __launch_bounds__(64) blend_bwd_mfma_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, int W,
    int H, int tiles_x, int tiles, const float* __restrict__ bg, int fc, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ grad_color,
    const float* __restrict__ grad_buffer, float* __restrict__ rows, uint8_t* __restrict__ row_valid) {
    constexpr int NV = ROW_FEAT + FC;
    constexpr int ROWF = ((NV + 3) / 4) * 4;
    constexpr int NC = 3 + FC;  // colour + feature columns of the W x Ggrad product
    constexpr int KK = (NC + 3) / 4;  // k-steps of the colour . gradient product (4 channels each) = channel quads
    constexpr int NQ = 3 + KK;        // record quads staged: geo0, geo1, bin, channels
    // +1 quad of padding per array: the 8 lanes that stage one record write quads q = 0..6 of the same
    // row; without the pad their addresses differ by a multiple of 128 B (7-way bank conflict)
    __shared__ float4 s_v[NQ][BB + 1];
    __shared__ uint32_t s_gid[BB];
    __shared__ uint32_t s_slot[BB];
    __shared__ uint32_t s_list[BB];
    __shared__ __align__(16) float s_g[64][16];    // per pixel: dL/dcolour (3), dL/dfeature (FC), zero pad
    __shared__ float4 s_px[64];                    // per pixel: running T, running Sg, n_contrib (bits), -
    __shared__ __align__(16) float s_d[16][16];    // geometry moments of the current group
    __shared__ uint32_t s_slotg[16];               // emission slot of each survivor of the current group

    const int b = blockIdx.x;
    const int tile = (b >> 5) * 8 + (b & 7);
    const int quad = (b >> 3) & 3;
    if (tile >= tiles) return;
    const int lane = threadIdx.x;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int qx0 = tile_x * GS2M_TILE + (quad & 1) * 8, qy0 = tile_y * GS2M_TILE + (quad >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;
    const float bx0 = (float)qx0, bx1 = bx0 + 7.0f, by0 = (float)qy0, by1 = by0 + 7.0f;
    const uint2 range = ranges[tile];

    // ---- pixel-per-lane prologue: lane = pixel (lx = lane & 7, ly = lane >> 3) ----
    uint32_t lastp;
    {
        const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
        const bool inside = px < W && py < H;
        const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
        const float Tf = inside ? final_T[pix] : 0.f;
        lastp = inside ? n_contrib[pix] : 0u;
        float g[16];
#pragma unroll
        for (int k = 0; k < 16; k++) g[k] = 0.f;
        if (inside) {
            g[0] = grad_color[pix]; g[1] = grad_color[HW + pix]; g[2] = grad_color[2 * HW + pix];
#pragma unroll
            for (int ch = 0; ch < FC; ch++) g[3 + ch] = ch < fc ? grad_buffer[ch * HW + pix] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4*>(&s_g[lane][4 * q]) = make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
        // suffix sum seeded with the background term (backward.cu:562-566)
        s_px[lane] = make_float4(Tf, Tf * (bg[0] * g[0] + bg[1] * g[1] + bg[2] * g[2]), u2f(lastp), 0.f);
    }
    uint32_t m = lastp;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    const int maxc = (int)m;  // entries past the quadrant's largest n_contrib are never touched
    const int nb = (maxc + BB - 1) / BB;
    gs2m_sync();

    // ---- survivor-per-lane state ----
    const int j = lane & 15, r = lane >> 4;
    // The running per-pixel transmittance / suffix sum live in LDS (s_px): read by the 16 survivor lanes
    // of the pixel's row each step, written back by its last lane.  LDS operations of one wave execute
    // in order, so the next group's read sees this group's write.
    const float qxr = (float)(qx0 + r), qyf = (float)qy0;
    const float ph0 = j == 0 ? 1.f : 0.f, ph1 = j == 1 ? 1.f : 0.f, ph2 = j == 2 ? 1.f : 0.f, ph3 = j == 3 ? 1.f : 0.f,
                ph4 = j == 4 ? 1.f : 0.f, ph5 = j == 5 ? 1.f : 0.f;  // one-hot selector of this lane's moment column
    const float halfW = 0.5f * W, halfH = 0.5f * H;
    const float xq = (float)qx0 + 3.5f, yq = (float)qy0 + 3.5f;

    float sx = 0.f, sy = 0.f, sA = 0.f, sB = 0.f, sC = 0.f, so = 0.f;
    // colour . gradient dot products gc[survivor][pixel] come from the matrix pipe as well:
    //   D[i][n] = sum_k A[i][k] B[k][n],  i = pixel slot of a 16-pixel block, n = survivor, k = channel.
    // Lane (j, r) receives D[4r + rr][j] in accumulator element rr, so with pixel slot 4r + rr := the pixel
    // this lane evaluates in step 4b + rr (p = 16b + 4rr + r) the four elements are exactly the four
    // steps' gc -- no transposition.  A[i][k] = s_g[16b + 4(i & 3) + (i >> 2)][4kk + k] (lane i = j, k = r),
    // B[k][n] = channel 4kk + r of survivor j: KK registers per lane instead of 3 + FC.
    float scB[KK];
#pragma unroll
    for (int k = 0; k < KK; k++) scB[k] = 0.f;
    // channel 4kk + r is element r of record quad REC_CH + kk (pad channels are 0 in the record and in s_g)
    const float* chB = reinterpret_cast<const float*>(&s_v[REC_CH][0]) + r;
    const float* gA = &s_g[4 * (j & 3) + (j >> 2)][r];  // + 16 b rows, + 4 kk columns
    uint32_t spos = 0xFFFFFFFFu;  // empty slot: behind every pixel's last contributor
    int nfill = 0;
    const float pxf0 = qxr, pxf1 = qxr + 4.0f;
    // Phi[p][j] = 1, cx, cy, cx^2, cx*cy, cy^2 (j = 0..5) as phA + cy * (phB + ph5 * cy); cx takes two values per lane
    const float cx0 = (float)r - 3.5f, cx1 = (float)r + 0.5f;
    const float phA0 = __builtin_fmaf(cx0, __builtin_fmaf(ph3, cx0, ph1), ph0), phA1 = __builtin_fmaf(cx1, __builtin_fmaf(ph3, cx1, ph1), ph0);
    const float phB0 = __builtin_fmaf(ph4, cx0, ph2), phB1 = __builtin_fmaf(ph4, cx1, ph2);

    // one group = up to 16 survivors: 16 steps of (16 survivors x 4 pixels), then the epilogue
    auto process_group = [&](int nvalid) {
        v4f acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        float U1 = 0.f, U2 = 0.f;  // per-lane partial |.| sums over this lane's 16 pixels
        auto gc_block = [&](int b) {
            v4f a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KK; k++) a = __builtin_amdgcn_mfma_f32_16x16x4f32(gA[(16 * b) * 16 + 4 * k], scB[k], a, 0, 0, 0);
            return a;
        };
        v4f gnext = gc_block(0);
#pragma unroll GS2M_BWDM_UNROLL_B
        for (int b = 0; b < 4; b++) {
            const v4f gcur = gnext;
            const float pyb = qyf + (float)(2 * b), cyb = (float)(2 * b) - 3.5f;
            // the block's LDS operands up front: the compiler cannot move these reads above the s_px writes of
            // earlier steps on its own (it cannot see that the four pixels differ), and every step would wait
            // out a full LDS latency twice
            float4 pstv[4];
            float gBv[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                pstv[rr] = s_px[16 * b + 4 * rr + r];
                gBv[rr] = s_g[16 * b + 4 * rr + r][j];
            }
            float gAn[KK];  // A operand of the NEXT block's colour . gradient product
#pragma unroll
            for (int k = 0; k < KK; k++) gAn[k] = gA[(16 * ((b + 1) & 3)) * 16 + 4 * k];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int p = 16 * b + 4 * rr + r;
                const float pxf = (rr & 1) ? pxf1 : pxf0, pyf = (rr & 2) ? pyb + 1.0f : pyb;
                const float dx = sx - pxf, dy = sy - pyf;
                const float power = gs2m_power(dx, dy, sA, sB, sC);
                const float G = gs2m_exp(power);
                const float alpha = fminf(0.99f, so * G);
                const float4 pst = pstv[rr];  // running T, running Sg, n_contrib
                const bool contrib = (spos <= f2u(pst.z)) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
                const float am = contrib ? alpha : 0.f;
                const float Gm = contrib ? G : 0.f;
                const float inv = __builtin_amdgcn_rcpf(1.f - am);
                const float Pinc = row_scan_mul(inv);
                const float Ti = pst.x * Pinc;  // transmittance in front of survivor j at this pixel
                const float w = am * Ti;
                const float gc = gcur[rr];
                const float qv = gc * w;
                const float Sinc = row_scan_add(qv);
                const float Sprev = pst.y + (Sinc - qv);  // contributions of everything behind survivor j
                const float da = Ti * gc - Sprev * inv;   // dL/dalpha (header of blend_bwd.hip)
                if (j == 15) *reinterpret_cast<float2*>(&s_px[p]) = make_float2(Ti, pst.y + Sinc);
                const float s = so * da * Gm;
                const float t1 = dx * sA + dy * sB, t2 = dy * sC + dx * sB;
                U1 += fabsf(s * t1);
                U2 += fabsf(s * t2);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, gBv[rr], acc1, 0, 0, 0);
                const float cy = (rr & 2) ? cyb + 1.0f : cyb;
                const float phi = __builtin_fmaf(cy, __builtin_fmaf(ph5, cy, (rr & 1) ? phB1 : phB0), (rr & 1) ? phA1 : phA0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(s, phi, acc2, 0, 0, 0);
                if (rr == 1) {  // one block ahead, operands long since loaded: the matrix pipe's latency stays hidden
                    v4f a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < KK; k++) a = __builtin_amdgcn_mfma_f32_16x16x4f32(gAn[k], scB[k], a, 0, 0, 0);
                    gnext = a;
                }
            }
        }
        // |.| sums: add the 4 pixel rows of each survivor (lanes j, j+16, j+32, j+48)
        {
            const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(U1), __float_as_uint(U2), false, false);
            float h = __uint_as_float(a[0]) + __uint_as_float(a[1]);  // lanes 0-31: U1 halves, lanes 32-63: U2 halves
            const auto bsw = __builtin_amdgcn_permlane16_swap(__float_as_uint(h), __float_as_uint(h), false, false);
            h = __uint_as_float(bsw[0]) + __uint_as_float(bsw[1]);    // row 0 (and 1): total U1, row 2 (and 3): total U2
            U1 = h;
        }
        // ---- epilogue: lane (j, r) holds D[4r + rr][j], rr = 0..3 ----
        gs2m_sync();
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
            if (j < 6) s_d[4 * r + rr][j] = acc2[rr];
        if (r == 0) s_d[j][6] = U1;
        if (r == 2) s_d[j][7] = U1;
        gs2m_sync();
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {  // colour / feature sums: 16 consecutive lanes own one survivor's row
            const int i = 4 * r + rr;
            if (i < nvalid && j < ROWF - ROW_COL) {
                const size_t rslot = (size_t)s_slotg[i] * 4 + quad;
                rows[rslot * ROWF + ROW_COL + j] = j < NC ? acc1[rr] : 0.f;
            }
        }
        if (r == 0 && j < nvalid) {  // geometry sums of survivor j from its moments
            const float4 m0 = *reinterpret_cast<const float4*>(&s_d[j][0]);  // M0, Mx, My, Mxx
            const float4 m1 = *reinterpret_cast<const float4*>(&s_d[j][4]);  // Mxy, Myy, U1, U2
            const float xc = sx - xq, yc = sy - yq;  // dx = xc - cx, dy = yc - cy
            const float Sdx = xc * m0.x - m0.y, Sdy = yc * m0.x - m0.z;
            const float Sdxx = xc * xc * m0.x - 2.f * xc * m0.y + m0.w;
            const float Sdxy = xc * yc * m0.x - xc * m0.z - yc * m0.y + m1.x;
            const float Sdyy = yc * yc * m0.x - 2.f * yc * m0.z + m1.y;
            const size_t rslot = (size_t)s_slotg[j] * 4 + quad;
            float4* o4 = reinterpret_cast<float4*>(rows + rslot * ROWF);
            o4[0] = make_float4(-halfW * (sA * Sdx + sB * Sdy), -halfH * (sC * Sdy + sB * Sdx), halfW * m1.z, halfH * m1.w);
            o4[1] = make_float4(-0.5f * Sdxx, -0.5f * Sdxy, -0.5f * Sdyy, m0.x != 0.f ? m0.x / so : 0.f);
            row_valid[rslot] = 1;
        }
    };

    // Staging is two dependent memory accesses (list entry -> record) that nothing inside the wave overlaps.
    // The list entries are therefore fetched one batch ahead into a register, and the next batch's record
    // lines are touched (one dword each, result unused) before this batch's groups are processed: when the
    // real staging loads come they hit L2.  Costs 2 VGPRs instead of a second staging buffer.
    uint32_t gid_next = 0, touch = 0;
    if (nb > 0 && lane < min(BB, maxc - (nb - 1) * BB)) gid_next = point_list[range.x + (nb - 1) * BB + lane];
    for (int bi = nb - 1; bi >= 0; bi--) {
        const int base = bi * BB;
        const int cnt = min(BB, maxc - base);
        asm volatile("" ::"v"(touch));  // the touch loads of the previous iteration retire here at the latest
        gs2m_sync();
        if (lane < cnt) s_gid[lane] = gid_next;
        if (bi > 0 && lane < BB) gid_next = point_list[range.x + base - BB + lane];
        gs2m_sync();
        {
            const int q = lane & 7;
            if (q < NQ) {
#pragma unroll
                for (int rr = 0; rr < BB / 8; rr++) {
                    const int row = rr * 8 + (lane >> 3);
                    if (row < cnt) {
                        const float4 v = rec[(size_t)s_gid[row] * REC_Q + q];
                        if (q == REC_BIN) {
                            const uint32_t off = f2u(v.x), rm = f2u(v.y), rw = f2u(v.z) & 0xFFFFu;
                            s_slot[row] = off + ((uint32_t)tile_y - (rm >> 16)) * rw + ((uint32_t)tile_x - (rm & 0xFFFFu));
                        }
                        s_v[q][row] = v;
                    }
                }
            }
        }
        gs2m_sync();
        bool hit = false;
        if (lane < cnt) {
            const float4 a = s_v[REC_GEO0][lane], c = s_v[REC_GEO1][lane];
            hit = gs2m_reaches_rect(a.x, a.y, a.z, a.w, c.x, c.z, c.w, s_v[REC_BIN][lane].w, bx0, bx1, by0, by1);
        }
        const unsigned long long mask = __ballot(hit);
        // back to front: the hit with the highest list position gets rank 0
        if (hit) s_list[lane == 63 ? 0 : (int)__popcll(mask >> (lane + 1))] = (uint32_t)lane;
        const int nh = (int)__popcll(mask);
        if (bi > 0 && lane < BB) touch = *reinterpret_cast<const volatile uint32_t*>(rec + (size_t)gid_next * REC_Q);
        gs2m_sync();
        int taken = 0;
        while (taken < nh) {
            const int n = min(16 - nfill, nh - taken);
            const int want = j - nfill;
            if (want >= 0 && want < n) {
                const int jj = (int)s_list[taken + want];
                const float4 a = s_v[REC_GEO0][jj], c = s_v[REC_GEO1][jj];
                sx = a.x; sy = a.y; sA = a.z; sB = a.w; sC = c.x; so = c.y;
#pragma unroll
                for (int k = 0; k < KK; k++)
                    scB[k] = chB[(k * (BB + 1) + jj) * 4];
                spos = (uint32_t)(base + jj + 1);
                if (r == 0) s_slotg[j] = s_slot[jj];
            }
            nfill += n;
            taken += n;
            if (nfill == 16) {
                process_group(16);
                nfill = 0;
                spos = 0xFFFFFFFFu;
            }
        }
    }
    if (nfill > 0) process_group(nfill);
}

int fc_template(int fc) { return fc <= 1 ? 1 : (fc <= 5 ? 5 : (fc <= 9 ? 9 : 10)); }

}  // namespace

int gs2m_row_floats_mfma(int fc) { return ((ROW_FEAT + fc_template(fc) + 3) / 4) * 4; }

void gs2m_launch_blend_bwd_mfma(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                                const BinningState& b, const ImageState& im, const float* grad_color,
                                const float* grad_buffer, float* rows, uint8_t* row_valid, hipStream_t s) {
    const int tiles = tiles_x * tiles_y;
    const int grid = ((tiles + 7) / 8) * 32;
#define GS2M_BWDM(FC)                                                                                                      \
    blend_bwd_mfma_kernel<FC><<<grid, 64, 0, s>>>(im.ranges, b.point_list, g.rec, W, H, tiles_x, tiles, bg, fc, im.final_T, \
                                                  im.n_contrib, grad_color, grad_buffer, rows, row_valid)
    switch (fc_template(fc)) {
        case 1: GS2M_BWDM(1); break;
        case 5: GS2M_BWDM(5); break;
        case 9: GS2M_BWDM(9); break;
        default: GS2M_BWDM(10); break;
    }
#undef GS2M_BWDM
}
