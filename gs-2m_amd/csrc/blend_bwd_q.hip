// Backward of the alpha compositing for gfx950.
//
// Semantics: renderCUDA (bwd), diff-gaussian-rasterization/cuda_rasterizer/backward.cu:413-598 -- same per-pixel tests,
// same gradients.  The reference walks a tile's list back to front with one thread per pixel and adds every
// (pixel, Gaussian) pair's 11 + fc gradient components to global memory with atomicAdd (:551-595).  Here:
//  * a wave owns one 8x8 quadrant and walks the quadrant's own list (tile_sort.hip) from the
//    quadrant's last contributor backwards, 16 entries ("survivors": every entry passed the quadrant test when the list
//    was built) per group;
//  * lane = (survivor j = lane & 15, pixel column r = lane >> 4): a group walks the 64 pixels in 8 double-steps, each
//    lane evaluating its two pixels (r, y) and (r + 4, y) of one image row as ONE packed fp32 pair;
//  * the reference's per-channel "colour behind" recurrence (backward.cu:530-550: accum_rec = alpha_k c_k + (1 - alpha_k)
//    accum_rec, dL/dalpha += (c - accum_rec) dL/dpixel) is projected on the pixel's gradient g: with b_i = (colour behind
//    survivor i) . g, seeded with bg . g (backward.cu:562-566), dL/dalpha_i = T_i ((c_i . g) - b_i) and b obeys the AFFINE
//    recurrence b_i = (1 - alpha_k) b_k + alpha_k (c_k . g), k the survivor right behind i: ONE scalar recurrence per pixel
//    instead of one per channel, a convex blend like the reference's (its rounding errors do not accumulate along the
//    list; rounds 1-5 carried the suffix SUM S_i = sum w_k (c_k . g) and formed T_i (c_i . g) - S_i / (1 - alpha_i): the same
//    number as the difference of two running quantities, 6-10 x the reference's error in the far tail).  The maps of the 16
//    survivors of a group are composed by a Kogge-Stone scan over the 16 lanes of a DPP row (row_scan_affine2); the product
//    part of the scanned map is prod (1 - alpha): T_i = (transmittance behind the group) / that product -- one reciprocal
//    per pair, of the scanned product (the reference divides entry by entry, backward.cu:532); the values at the group's
//    front are carried to the next group through LDS (s_T2, s_B2);
//  * the sums over pixels are fp32 MFMAs (v_mfma_f32_16x16x4_f32): dL/dcolour,feature[survivor][channel] =
//    W[survivor][pixel] x Ggrad[pixel][channel], and the colour . gradient dots (c_i . g)[pixel][survivor] =
//    Ggrad x C^T in an accumulator layout that needs no transposition (see scB / gA below);
//  * the geometry gradients (dL/dmean2D with its two |.| channels, dL/dconic, dL/dopacity) come from six moments of
//    s = opacity dL/dalpha G about the quadrant centre, reduced over the 4 pixel-column lanes of a survivor with
//    permlane32 / permlane16 swaps;
//  * lane (j, r) loads the geometry and its channel of entry `top - j` straight from the 128-B record (L2) and the entry's
//    gradient row from the list (round 5: it used to be two dependent gathers through the record's binning quad); the loads
//    for the next group are issued before the current group's epilogue and land behind it.
// One wave per quadrant (64-thread workgroups, no barriers), XCD-aware block ids, one partial-gradient row per
// (instance, quadrant) in the numbering emit_kernel / emit_heavy_kernel prepared (binning.hip): the rows of a Gaussian are one
// contiguous run, every row is written (zeros for the entries behind a quadrant's last contributor), and
// gaussian_bwd.hip streams them.  No float atomics anywhere: gradients are bitwise reproducible.
// Measured and not kept in round 3: one workgroup per TILE whose four quadrant waves combine an
// instance's sums in an LDS table before they leave the chip (one row per instance instead of one per quadrant).
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
// floats per partial-gradient row: 11 + fc padded to whole float4s (80 B at fc 9).  GS2M_ROW_SECTORS (A/B builds, profiles/r06_rows_ab.md):
// whole 32-byte sectors (96 B at fc 9) -- no partial-sector writes, 20 % more bytes both ways.
#ifdef GS2M_ROW_SECTORS
#define GS2M_ROW_ROUND(nv) ((((nv) + 7) / 8) * 8)
#else
#define GS2M_ROW_ROUND(nv) ((((nv) + 3) / 4) * 4)
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef GS2M_BWDQ_WAVES
#define GS2M_BWDQ_WAVES
#endif
#ifndef GS2M_BWDQ_UNROLL_B
#define GS2M_BWDQ_UNROLL_B 1
#endif

// Inclusive scan of AFFINE maps b -> A b + B over the 16 lanes of a DPP row (Kogge-Stone), for a PAIR of independent pixels with
// the two chains interleaved.  Lane j holds survivor j's map (A = 1 - alpha, B = alpha (c . g)); afterwards it holds the
// composition of the maps of lanes 0..j, lane 0's applied first: A = prod (1 - alpha), B = the colour . gradient blended over
// survivors 0..j with nothing behind them.  One level with stride d: (A, B) <- (A A[-d], A B[-d] + B) -- one v_fmac_f32_dpp
// (B first: it reads this lane's A of the level before) and one v_mul_f32_dpp per chain; lanes whose source falls outside the
// row are disabled by the DPP and keep their map (the identity composed in front).  A DPP read needs 2 wait states after the
// VALU write of its source: with two chains of two registers every register is read three instructions after it was written,
// so only the first level needs the `s_nop` (the compiler does not track the hazard through inline asm).
__device__ __forceinline__ void row_scan_affine2(float& Ax, float& Ay, float& Bx, float& By) {
#define GS2M_AFFINE_LEVEL(D)                                                             \
        "v_fmac_f32_dpp %2, %2, %0 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t"          \
        "v_fmac_f32_dpp %3, %3, %1 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t"          \
        "v_mul_f32_dpp %0, %0, %0 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t"           \
        "v_mul_f32_dpp %1, %1, %1 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t"
        GS2M_AFFINE_LEVEL(1)
        GS2M_AFFINE_LEVEL(2)
        GS2M_AFFINE_LEVEL(4)
        GS2M_AFFINE_LEVEL(8)
        : "+v"(Ax), "+v"(Ay), "+v"(Bx), "+v"(By));
#undef GS2M_AFFINE_LEVEL
}
// value of the lane below in the DPP row; lane 0 of the row takes `first`
__device__ __forceinline__ float row_shr1(float first, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x111 /* row_shr:1 */, 0xF, 0xF, false));
}

template <int FC>
__global__ void __launch_bounds__(64) GS2M_BWDQ_WAVES blend_bwd_q_kernel(
    const uint2* __restrict__ ranges, const uint2* __restrict__ qlist, const uint32_t* __restrict__ qlast,
    const uint32_t* __restrict__ qcount, const uint32_t* __restrict__ qrow,
    const float4* __restrict__ rec, int W, int H, int tiles_x, int tiles, const float* __restrict__ bg, int fc,
    const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ grad_color,
    const float* __restrict__ grad_buffer, float* __restrict__ rows) {
    constexpr int NV = ROW_FEAT + FC;
    constexpr int ROWF = GS2M_ROW_ROUND(NV);
    constexpr int RSTRIDE = ROWF;  // rows are packed (a 128-B stride was tried: random single lines read no faster)
    constexpr int NC = 3 + FC;  // colour + feature columns of the W x Ggrad product
    constexpr int KK = (NC + 3) / 4;  // k-steps of the colour . gradient product (4 channels each) = channel quads
#ifndef GS2M_BWDQ_GSTRIDE
#define GS2M_BWDQ_GSTRIDE 20  // 16 + 4: with a 16-float stride the A-operand reads (16 pixel rows x 4 columns) fall on 16 banks
#endif
    constexpr int GST = GS2M_BWDQ_GSTRIDE;
    __shared__ __align__(16) float s_g[64][GST];   // per pixel: dL/dcolour (3), dL/dfeature (FC), zero pad
    // per pixel pair {(x, y), (x + 4, y)}, index 4 y + (x & 3): running T, running (colour behind) . g, n_contrib
    __shared__ float2 s_T2[32];
    __shared__ float2 s_B2[32];
    __shared__ uint2 s_N2[32];
    __shared__ __align__(16) float s_out[16][ROWF];  // the group's 16 gradient rows, assembled here and stored as whole float4s
    __shared__ uint32_t s_rowg[16];                  // gradient-row index of each survivor of the current group

    const int b = blockIdx.x;
    const int tile = (b >> 5) * 8 + (b & 7);
    const int quad = (b >> 3) & 3;
    if (tile >= tiles) return;
    const int lane = threadIdx.x;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int qx0 = tile_x * GS2M_TILE + (quad & 1) * 8, qy0 = tile_y * GS2M_TILE + (quad >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;
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
        // behind everything: the background (backward.cu:562-566)
        const int pe = ((lane >> 3) * 4 + (lane & 3)) * 2 + ((lane >> 2) & 1);  // pair index * 2 + element
        reinterpret_cast<float*>(s_T2)[pe] = Tf;
        reinterpret_cast<float*>(s_B2)[pe] = bg[0] * g[0] + bg[1] * g[1] + bg[2] * g[2];
        reinterpret_cast<uint32_t*>(s_N2)[pe] = lastp;
    }
    gs2m_sync();

    // ---- survivor-per-lane state ----
    const int j = lane & 15, r = lane >> 4;
    // The running per-pixel transmittance / colour . gradient behind live in LDS (s_T2, s_B2): read by the 16 survivor lanes
    // of the pixel's row each step, written back by its last lane.  LDS operations of one wave execute
    // in order, so the next group's read sees this group's write.
    const float qxr = (float)(qx0 + r), qyf = (float)qy0;
    const float halfW = 0.5f * W, halfH = 0.5f * H;

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
    // channel 4kk + r is float 4 (REC_CH + kk) + r of the record (pad channels are 0 in the record and in s_g)
    const float* gA = &s_g[4 * (j & 3) + (j >> 2)][r];  // + 16 b rows, + 4 kk columns
    uint32_t spos = 0xFFFFFFFFu;  // empty slot: behind every pixel's last contributor
    const v2f pxf2 = {qxr, qxr + 4.0f};                      // this lane's two pixel columns

    // The quadrant's list, from its last contributor backwards: group g holds entries top - 16 g - j, j = 0..15
    // (j = 0 is the back-most).  `np` = entries up to and including the last one that contributes to any pixel of
    // the quadrant (written by the forward); later entries are never touched (backward.cu:493-494, 519-520).
    const uint32_t len = range.y - range.x;
    const uint2* list = qlist + (size_t)4 * range.x + (size_t)quad * len;
    const uint32_t* lrow = qrow + (size_t)4 * range.x + (size_t)quad * len;  // gradient row of every list entry (tile_sort.hip)
    const int np = (int)qlast[tile * 4 + quad];
    const int ngroups = (np + 15) >> 4;
    // what a lane takes from its survivor's list entry and record: geometry, its channel of every quad, the gradient row
    struct Fill {
        float4 g0;       // x, y, A, B
        float2 g1;       // C, opacity
        float ch[KK];
        uint32_t pos1;   // position in the tile list + 1
        uint32_t row;    // gradient row of (instance, quadrant): numbered per Gaussian (binning.hip: emit_kernel, emit_heavy_kernel)
    };
    struct Entry {
        uint2 e;
        uint32_t row;
    };
    // list entry of survivor j in group g.  Lanes past the front of the list (the last group may be partial) take
    // entry 0 with position ~0 = behind every pixel's last contributor: real, finite record data that no pixel accepts.
    auto load_entry = [&](int g) {
        const int p = np - 1 - (16 * g + j);
        Entry en;
        en.e = list[max(p, 0)];
        en.row = lrow[max(p, 0)];
        en.e.y = p >= 0 ? en.e.y + 1u : 0xFFFFFFFFu;  // position in the tile list + 1
        return en;
    };
    auto load_fill = [&](const Entry en) {
        Fill f;
        const float4* p = rec + (size_t)(en.e.x & GS2M_GID_MASK) * REC_Q;
        f.g0 = p[REC_GEO0];
        f.g1 = *reinterpret_cast<const float2*>(p + REC_GEO1);
#pragma unroll
        for (int k = 0; k < KK; k++) f.ch[k] = reinterpret_cast<const float*>(p + REC_CH + k)[r];
        f.pos1 = en.e.y;
        f.row = en.row;
        return f;
    };
    // start of gradient row `row`: the row count is far below 2^32 / 6, so row * (floats per row / 4) is formed in 32 bits
    // (one shift-add) and only widened for the shift by 16 bytes -- `rows + (size_t)row * RSTRIDE` compiles to a
    // quarter-rate 64-bit multiply-add, five of them per group epilogue
    auto row_ptr = [&](uint32_t row) -> float* {
        return reinterpret_cast<float*>(reinterpret_cast<float4*>(rows) + (size_t)(row * (uint32_t)(RSTRIDE / 4)));
    };
    Fill f;
    uint32_t row_cur = 0;  // gradient row of this lane's survivor in the current group
    Entry e_next = {make_uint2(0u, 0u), 0u};
    int g_cur = 0;
    // the next group's record values are requested between this group's steps and its epilogue and land behind the
    // epilogue; the list entries one group further ahead
    auto issue_next = [&]() {
        if (g_cur + 1 < ngroups) f = load_fill(e_next);
        if (g_cur + 2 < ngroups) e_next = load_entry(g_cur + 2);
    };
    // the rows of the previous group, parked in s_out by its epilogue: 16 x ROWF / 4 float4, one per lane and round
    int pending = 0;  // survivors of the previous group whose rows are still in LDS (wave-uniform)
    auto flush_rows = [&]() {
        constexpr int RQ = ROWF / 4;  // float4 per row
#pragma unroll
        for (int t0 = 0; t0 < 16 * RQ; t0 += GS2M_WAVE) {
            const int t = t0 + lane, e = t / RQ, q = t - e * RQ;
            if (t < 16 * RQ && e < pending)
                reinterpret_cast<float4*>(row_ptr(s_rowg[e]))[q] = *reinterpret_cast<const float4*>(&s_out[e][4 * q]);
        }
        pending = 0;
    };
    // one group = up to 16 survivors: 16 steps of (16 survivors x 4 pixels), then the epilogue
    auto process_group = [&](int nvalid) {
        v4f acc1 = {0.f, 0.f, 0.f, 0.f};
        float U1 = 0.f, U2 = 0.f;  // per-lane partial |.| sums over this lane's 16 pixels
        // per-lane moments of s over its 16 pixels, per pixel column, about the survivor's OWN mean (dy = mean.y - pixel.y):
        // sum s, sum s dy, sum s dy^2.  (Rounds 1-3 took them about the quadrant centre -- per-pixel constants -- and shifted
        // them to the mean afterwards: for splats centred hundreds of pixels outside the image that shift cancels by orders of
        // magnitude and had to be done in double; these sums are what the reference itself accumulates, backward.cu:571-592.)
        v2f m0 = {0.f, 0.f}, m1 = {0.f, 0.f}, m2 = {0.f, 0.f};
        auto gc_block = [&](int b) {
            v4f a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KK; k++) a = __builtin_amdgcn_mfma_f32_16x16x4f32(gA[(16 * b) * GST + 4 * k], scB[k], a, 0, 0, 0);
            return a;
        };
        v4f gnext = gc_block(0);
        const v2f sx2 = {sx, sx}, sA2 = {sA, sA}, sB2 = {sB, sB}, so2 = {so, so};
#pragma unroll GS2M_BWDQ_UNROLL_B
        for (int b = 0; b < 4; b++) {
            const v4f gcur = gnext;
            const float pyb = qyf + (float)(2 * b);
            // the block's LDS operands up front: the compiler cannot move these reads above the s_T2/s_B2 writes
            // of earlier steps on its own (it cannot see that the pixels differ), and every step would wait out
            // a full LDS latency twice
            v2f T2[2], B2[2];
            uint2 N2[2];
            float gBv[4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int pi = (2 * b + h) * 4 + r;
                const float2 t = s_T2[pi], q = s_B2[pi];
                T2[h] = v2f{t.x, t.y};
                B2[h] = v2f{q.x, q.y};
                N2[h] = s_N2[pi];
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++) gBv[rr] = s_g[16 * b + 4 * rr + r][j];
            float gAn[KK];  // A operand of the NEXT block's colour . gradient product
#pragma unroll
            // (the last block has no successor: it reads its own rows again -- an address select instead of a branch per operand)
            for (int k = 0; k < KK; k++) gAn[k] = gA[(16 * (b < 3 ? b + 1 : 3)) * GST + 4 * k];
#pragma unroll
            for (int h = 0; h < 2; h++) {  // image row 2b + h: pixels (r, 2b + h) and (r + 4, 2b + h) as one packed pair
                const int pi = (2 * b + h) * 4 + r;
                const float pyf = h ? pyb + 1.0f : pyb;
                const float dy = sy - pyf;
                const v2f dx = sx2 - pxf2;
                // gs2m_power's operation order (the forward's alpha must be reproduced bit for bit)
                const float cdy = sC * dy;
                const float t2 = cdy * dy;
                const v2f t1 = (sA2 * dx) * dx;
                const v2f t3 = (sB2 * dx) * dy;
                const v2f power = (-0.5f * (t1 + t2)) - t3;
                const v2f e = power * GS2M_LOG2E;
                const v2f G = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
                const v2f soG = so2 * G;  // alpha before the 0.99 clamp; alpha >= 1/255 <=> soG >= 1/255
                const bool c0 = (spos <= N2[h].x) && (power.x <= 0.0f) && (soG.x >= 1.0f / 255.0f);
                const bool c1 = (spos <= N2[h].y) && (power.y <= 0.0f) && (soG.y >= 1.0f / 255.0f);
                const v2f sg = {c0 ? soG.x : 0.f, c1 ? soG.y : 0.f};    // opacity * G of contributing pairs, else 0
                const v2f am = {fminf(0.99f, sg.x), fminf(0.99f, sg.y)};  // their alpha, else 0
                const v2f om = 1.0f - am;
                const v2f gc = {gcur[2 * h], gcur[2 * h + 1]};
                // survivor j's map of the colour-behind recurrence (backward.cu:530-550 projected on the pixel's gradient):
                // b -> (1 - alpha) b + alpha (c . g); a pair that does not contribute is the identity
                const v2f bq = am * gc;
                float Ax = om.x, Ay = om.y, Bx = bq.x, By = bq.y;
                row_scan_affine2(Ax, Ay, Bx, By);
                // (colour . gradient) behind survivor j + 1 = the scanned map applied to what lies behind the group; behind survivor j
                // itself: the lane below (plain fp32 instructions on the scan's four registers: as a packed pair they would have to be
                // moved into an aligned register pair first, and a packed instruction is no cheaper than two plain ones on gfx950)
#ifndef GS2M_BWDQ_PK_SCAN
                const v2f binc = {__builtin_fmaf(Ax, B2[h].x, Bx), __builtin_fmaf(Ay, B2[h].y, By)};
                const v2f bex = {row_shr1(B2[h].x, binc.x), row_shr1(B2[h].y, binc.y)};
                // transmittance in front of survivor j: what is left behind the group / prod (1 - alpha) over survivors 0..j
                // (backward.cu:532 divides once per entry; here ONE reciprocal of the scanned product)
                const v2f Ti = {T2[h].x * __builtin_amdgcn_rcpf(Ax), T2[h].y * __builtin_amdgcn_rcpf(Ay)};
#else
                const v2f Ainc = {Ax, Ay}, Binc = {Bx, By};
                const v2f binc = __builtin_elementwise_fma(Ainc, B2[h], Binc);
                const v2f bex = {row_shr1(B2[h].x, binc.x), row_shr1(B2[h].y, binc.y)};
                const v2f rP = {__builtin_amdgcn_rcpf(Ainc.x), __builtin_amdgcn_rcpf(Ainc.y)};
                const v2f Ti = T2[h] * rP;
#endif
                const v2f w = am * Ti;
                const v2f da = Ti * (gc - bex);  // dL/dalpha (header of this file)
                if (j == 15) {
                    s_T2[pi] = make_float2(Ti.x, Ti.y);
                    s_B2[pi] = make_float2(binc.x, binc.y);
                }
                const v2f sv = da * sg;  // s = opacity * dL/dalpha * G
                const v2f u1 = dx * sA2 + dy * sB, u2 = cdy + dx * sB2;
                const v2f a1 = sv * u1, a2 = sv * u2;
                U1 += fabsf(a1.x); U1 += fabsf(a1.y);
                U2 += fabsf(a2.x); U2 += fabsf(a2.y);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, gBv[2 * h], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, gBv[2 * h + 1], acc1, 0, 0, 0);
                m0 += sv;
                m1 = __builtin_elementwise_fma(sv, v2f{dy, dy}, m1);
                m2 = __builtin_elementwise_fma(sv, v2f{dy * dy, dy * dy}, m2);
                if (h == 0 && b < 3) {  // one block ahead, operands long since loaded: the result is there when the next block starts
                    v4f a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < KK; k++) a = __builtin_amdgcn_mfma_f32_16x16x4f32(gAn[k], scB[k], a, 0, 0, 0);
                    gnext = a;
                }
            }
            if (b == 0 && pending) flush_rows();  // the previous group's rows: their LDS writes are a whole block old by now
        }
        issue_next();
        // ---- per-survivor totals: add the 4 pixel rows of each survivor (lanes j, j+16, j+32, j+48) ----
        // two values at a time: permlane32_swap + add leaves value A's two half sums in lanes 0-31 and value B's in
        // lanes 32-63; permlane16_swap + add on two such registers leaves the totals of (A, C, B, D) in rows 0..3
        auto reduce4 = [&](float va, float vb, float vc, float vd) {
            const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
            const float hab = __uint_as_float(x[0]) + __uint_as_float(x[1]);
            const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(vc), __float_as_uint(vd), false, false);
            const float hcd = __uint_as_float(y[0]) + __uint_as_float(y[1]);
            const auto z = __builtin_amdgcn_permlane16_swap(__float_as_uint(hab), __float_as_uint(hcd), false, false);
            return __uint_as_float(z[0]) + __uint_as_float(z[1]);  // row 0: A, row 1: C, row 2: B, row 3: D
        };
        // sums over this lane's 16 pixels: s, s dx, s dy, s dx^2, s dx dy, s dy^2 (dx is fixed per pixel column)
        const v2f dxc = sx2 - pxf2;
        const float M0 = m0.x + m0.y, Sx = dxc.x * m0.x + dxc.y * m0.y, Sy = m1.x + m1.y;
        const float Sxx = (dxc.x * dxc.x) * m0.x + (dxc.y * dxc.y) * m0.y, Sxy = dxc.x * m1.x + dxc.y * m1.y, Syy = m2.x + m2.y;
        // The row's eight geometry entries (backward.cu:571-592) are linear in those sums: every lane scales its OWN partial
        // sums and the four pixel-column lanes of a survivor are added afterwards, so the totals land already final, entry r
        // and 4 + r of the row in lane (j, r) -- no second pass by a quarter of the lanes over exchanged sums.
        const float o0 = -halfW * (sA * Sx + sB * Sy), o1 = -halfH * (sC * Sy + sB * Sx), o2 = halfW * U1, o3 = halfH * U2;
        const float o7 = so > 0.f ? M0 * __builtin_amdgcn_rcpf(so) : 0.f;  // sum G dL/dalpha (v_rcp: 1 ulp); opacity 0 contributes nowhere
        const float R1 = reduce4(o0, o2, o1, o3);                               // rows: o0, o1, o2, o3
        const float R2 = reduce4(-0.5f * Sxx, -0.5f * Syy, -0.5f * Sxy, o7);  // rows: cxx, cxy, cyy, dopacity
        // The group's 16 rows are assembled in LDS and leave as whole float4s, 16 bytes per lane (rounds 1-3: four 4-byte
        // stores per lane for the colour part, two 16-byte stores from a quarter of the lanes for the geometry, each with its
        // own 64-bit address arithmetic).  One wave, in-order LDS: flush_rows() -- called from inside the NEXT group's steps,
        // so that nothing waits for the LDS round trip -- reads what is written here.
        s_out[j][r] = R1;
        s_out[j][4 + r] = R2;
        if (r == 0) s_rowg[j] = row_cur;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)  // lane (j, r) holds the colour / feature sums D[4r + rr][j]
            if (j < ROWF - ROW_COL) s_out[4 * r + rr][ROW_COL + j] = j < NC ? acc1[rr] : 0.f;
        pending = nvalid;
    };

    // Entries behind the quadrant's last contributor are never processed (backward.cu:493-494) but own a row each:
    // zeros, so that the per-Gaussian sum can stream every row of its range without a validity map.
    {
        const int n = (int)qcount[tile * 4 + quad];
        for (int i = np + lane; i < n; i += GS2M_WAVE) {
            float4* o4 = reinterpret_cast<float4*>(row_ptr(lrow[i]));
#pragma unroll
            for (int q4 = 0; q4 < ROWF / 4; q4++) o4[q4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (ngroups > 0) {
        f = load_fill(load_entry(0));
        e_next = load_entry(1);
        for (g_cur = 0; g_cur < ngroups; g_cur++) {
            // install the group's survivors
            sx = f.g0.x; sy = f.g0.y; sA = f.g0.z; sB = f.g0.w; sC = f.g1.x; so = f.g1.y;
#pragma unroll
            for (int k = 0; k < KK; k++) scB[k] = f.ch[k];
            spos = f.pos1;
            row_cur = f.row;
            process_group(min(16, np - 16 * g_cur));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flush_rows();  // the last group's
    }
}

int fc_template(int fc) { return fc <= 1 ? 1 : (fc <= 5 ? 5 : (fc <= 9 ? 9 : 10)); }

}  // namespace

int gs2m_row_floats(int fc) { return GS2M_ROW_ROUND(ROW_FEAT + fc_template(fc)); }

// every row of the dense numbering is written (zeros behind a quadrant's last contributor)
void gs2m_launch_blend_bwd_q(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                             const BinningState& b, const ImageState& im, const float* grad_color,
                             const float* grad_buffer, float* rows, hipStream_t s) {
    const int tiles = tiles_x * tiles_y;
    const int grid = ((tiles + 7) / 8) * 32;
#define GS2M_BWDQ(FC)                                                                                                      \
    blend_bwd_q_kernel<FC><<<grid, 64, 0, s>>>(im.ranges, b.qlist, im.qlast, im.qcount, b.qrow, g.rec, W, H, tiles_x, tiles, bg, fc, im.final_T, \
                                                  im.n_contrib, grad_color, grad_buffer, rows)
    switch (fc_template(fc)) {
        case 1: GS2M_BWDQ(1); break;
        case 5: GS2M_BWDQ(5); break;
        case 9: GS2M_BWDQ(9); break;
        default: GS2M_BWDQ(10); break;
    }
#undef GS2M_BWDQ
}
