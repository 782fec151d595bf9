// Per-quadrant front-to-back alpha compositing (forward) for gfx950.
//
// Semantics: renderCUDA, diff-gaussian-rasterization/cuda_rasterizer/forward.cu:246-372 -- the same per-pixel
// sequence of tests and updates, on the per-quadrant lists tile_sort.hip builds.
//
// One wave per 8x8 quadrant, pixel per lane, no workgroup barriers: every list entry survives the quadrant test by
// construction, so the wave walks its list straight down, 16 entries per chunk.  The chunk's blend records are
// gathered by the wave itself (lane = entry x record quad, full 16-B loads) a whole chunk ahead of their use and
// parked in LDS; the evaluation reads them back with wave-uniform (broadcast) addresses at constant offsets, so the
// inner loop is fully unrolled and spends no vector instruction on addressing, ballot walks or quadrant tests.
// Lane predicates live in SGPR pairs as wave masks.  A first version fetched the records with SCALAR loads
// (s_load_dwordx8 one entry ahead, SGPR operands): it ran at half the speed -- scalar loads return out of order, so
// the only wait is "all of them", the prefetch distance is one evaluation, and the ~0.45 us load latency was exposed
// on every entry.
// Two other forms were built and measured: the alpha of two consecutive entries as packed fp32
// pairs out of a component-wise LDS layout (5 fewer vector instructions per entry, but 76 VGPRs = 6 waves per SIMD
// and the LDS reads right in front of their use: 0.30 instead of 0.27 ms), and the scalar-load form with the record
// lines touched ahead by a vector load (no better: the scalar path itself is the latency).
// `observe` (forward.cu:348-350: one atomicAdd per pixel) is one integer atomic per (list entry, quadrant) that sees
// a pixel with T > 0.5, straight into the output tensor (zeroed by the preprocess kernel): only the first few
// entries of a list do, and the wave stops looking once no live pixel has T > 0.5.
#include "common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

template <int FC>  // feature channels blended (compile time); runtime fc <= FC
__global__ void __launch_bounds__(64) blend_fwd_q_kernel(
    const uint2* __restrict__ ranges, const uint2* __restrict__ qlist, const uint32_t* __restrict__ qcount,
    const float4* __restrict__ rec, int W, int H, int tiles_x, int tiles, const float* __restrict__ bg, int fc,
    float* __restrict__ out_color, float* __restrict__ out_buffer, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, int* __restrict__ observe, uint32_t* __restrict__ qlast) {
    constexpr int NC = 3 + FC;        // blended channels: r, g, b, features
    constexpr int KQ = (NC + 3) / 4;  // channel quads of the record
    constexpr int NP = (NC + 1) / 2;  // channel pairs accumulated

    const int b = blockIdx.x;
    const int tile = (b >> 5) * 8 + (b & 7);  // the four quadrants of a tile run on one XCD (block ids go round the 8 XCDs)
    const int quad = (b >> 3) & 3;
    if (tile >= tiles) return;
    const int lane = threadIdx.x;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int qx0 = tile_x * GS2M_TILE + (quad & 1) * 8, qy0 = tile_y * GS2M_TILE + (quad >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;

    const uint2 range = ranges[tile];
    const uint32_t len = range.y - range.x;
    const int n = __builtin_amdgcn_readfirstlane((int)qcount[tile * 4 + quad]);
    const uint2* list = qlist + (size_t)4 * range.x + (size_t)quad * len;

    // record quads parked per entry: geo0, geo1, channel quads (the bin quad is only needed on the rare observe path)
    constexpr int CH = 16;       // entries per chunk
    constexpr int NS = 2 + KQ;   // quads staged
    // one LDS block, [buffer][NS quads + the chunk's list entries {Gaussian id, position in the tile list + 1}][CH],
    // addressed from ONE per-buffer base kept in a VGPR the compiler cannot rematerialise: with separate arrays at
    // known addresses it re-creates two address registers from SGPRs for every entry (2 of 32 vector instructions)
    constexpr int BUFQ = NS * CH + CH / 2;  // float4 per buffer
    __shared__ float4 s_buf[2 * BUFQ];
    uint32_t lds_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lds_zero));  // opaque 0
    float4* const s_base = s_buf + lds_zero;
    auto q_at = [&](int buf, int slot, int jj) -> float4& { return s_base[buf * BUFQ + slot * CH + jj]; };
    auto e_at = [&](int buf, int jj) -> uint2& { return reinterpret_cast<uint2*>(s_base + buf * BUFQ + NS * CH)[jj]; };
    const int ej = lane & 15, eq = lane >> 4;  // staging role: entry ej of the chunk, quad slot eq (and eq + 4)
    auto quad_of = [](int slot) { return slot < 2 ? slot : slot + 1; };  // staged slot -> record quad (skips REC_BIN)

    float T = 1.0f;
    uint32_t last_contributor = 0;
    v2f acc[NP];  // acc[k] = channels 2k, 2k + 1
#pragma unroll
    for (int k = 0; k < NP; k++) acc[k] = v2f{0.f, 0.f};
    // Lane predicates are kept as wave masks in SGPR pairs and combined on the scalar unit; the selects take the
    // mask as v_cndmask's scalar operand.  (As `bool`s carried round the loop the compiler keeps them in VGPRs and
    // spends ~8 vector instructions per entry converting back and forth.)
    typedef unsigned long long mask_t;
    mask_t live = __builtin_amdgcn_ballot_w64(inside);  // pixels that have not finished (forward.cu:288, 340-343)
    bool watch = true;  // some live pixel may still have T > 0.5 (wave-uniform)
    int ilast = 0;      // list entries up to and including the last one some pixel accepted (wave-uniform)

    // one list entry (index i of the list, slot jj of LDS buffer `buf`): returns true when every pixel has finished
    auto eval = [&](const int i, const int buf, const int jj) {
        const float4 a = q_at(buf, 0, jj);
        const float2 b = *reinterpret_cast<const float2*>(&q_at(buf, 1, jj));
        float4 c[KQ];
#pragma unroll
        for (int q = 0; q < KQ; q++) c[q] = q_at(buf, 2 + q, jj);
        const uint2 ec = e_at(buf, jj);
        const float dx = a.x - pxf, dy = a.y - pyf;
        const float p2 = gs2m_power(dx, dy, a.z, a.w, b.x);
        const float alpha = fminf(0.99f, b.y * gs2m_exp(p2));
        const float test_T = T * (1.0f - alpha);
        const mask_t cand = live & __builtin_amdgcn_ballot_w64(p2 <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
        const mask_t fin = cand & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);  // these pixels stop here, without this entry
        const mask_t contrib = cand & ~fin;
        live &= ~fin;
        if (contrib != 0ull) ilast = i + 1;
        float tw;  // contrib ? T : 0 -- branch-free: non-contributing lanes add 0
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(tw) : "v"(T), "s"(contrib));
        const float w = alpha * tw;
        const v2f ww = {w, w};
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            if (4 * q < NC) acc[2 * q] = __builtin_elementwise_fma(v2f{c[q].x, c[q].y}, ww, acc[2 * q]);
            if (4 * q + 2 < NC) acc[2 * q + 1] = __builtin_elementwise_fma(v2f{c[q].z, c[q].w}, ww, acc[2 * q + 1]);
        }
        const uint32_t pos1 = ec.y;  // position in the tile list + 1 (parked that way), as the reference counts contributors
        asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(last_contributor) : "v"(pos1), "s"(contrib));
        if (watch) {
            const mask_t half = contrib & __builtin_amdgcn_ballot_w64(T > 0.5f);
            if (half != 0ull) {  // rare: only the front of a list
                if (lane == 0) atomicAdd(&observe[__builtin_amdgcn_readfirstlane(ec.x)], (int)__popcll(half));  // integer: the order does not matter
            }
            // T only falls: once no live pixel is above 0.5 nothing later can be
            watch = (live & __builtin_amdgcn_ballot_w64(T > 0.5f)) != 0ull;
        }
        asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(T) : "v"(test_T), "s"(contrib));
        return live == 0ull;
    };

    // Software pipeline by chunks of 16 entries: while chunk c is evaluated out of LDS buffer c & 1, the record quads
    // of chunk c + 1 are in flight towards registers (written to the other buffer after the evaluation) and the list
    // entries of chunk c + 2 are being fetched.  One wave, in-order LDS: no barriers.
    const int nchunks = (n + CH - 1) / CH;
    auto load_entry = [&](int c) {  // entry ej of chunk c (clamped inside the list: the surplus lanes are never evaluated)
        return list[min(c * CH + ej, n - 1)];
    };
    struct Stage { float4 a, b; };
    auto load_quads = [&](const uint2 e) {  // this lane's two quads of its entry's record
        Stage st;
        const float4* p = rec + (size_t)(e.x & GS2M_GID_MASK) * REC_Q;  // the entry carries the quadrant-hit mask above the id
        st.a = p[quad_of(eq)];
        st.b = eq + 4 < NS ? p[quad_of(eq + 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
        return st;
    };
    auto park = [&](int buf, const uint2 e, const Stage& st) {
        q_at(buf, eq, ej) = st.a;
        if (eq + 4 < NS) q_at(buf, eq + 4, ej) = st.b;
        if (eq == 0) e_at(buf, ej) = make_uint2(e.x & GS2M_GID_MASK, e.y + 1u);
    };
    if (nchunks > 0) {
        uint2 e1 = load_entry(0);
        park(0, e1, load_quads(e1));
        e1 = nchunks > 1 ? load_entry(1) : e1;
        bool stop = false;
        for (int c = 0; c < nchunks && !stop; c++) {
            Stage st;
            const uint2 ecur = e1;
            if (c + 1 < nchunks) st = load_quads(ecur);
            if (c + 2 < nchunks) e1 = load_entry(c + 2);
            const int base = c * CH, buf = c & 1;
#pragma unroll
            for (int jj = 0; jj < CH; jj++) {
                if (base + jj >= n) break;
                if (eval(base + jj, buf, jj)) { stop = true; break; }
            }
            if (c + 1 < nchunks && !stop) park(buf ^ 1, ecur, st);
        }
    }

    if (lane == 0) qlast[tile * 4 + quad] = (uint32_t)ilast;
    if (inside) {
        const size_t HW = (size_t)H * W;
        const size_t pix = (size_t)py * W + px;
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
        out_color[pix] = acc[0][0] + T * bg[0];
        out_color[HW + pix] = acc[0][1] + T * bg[1];
        out_color[2 * HW + pix] = acc[1][0] + T * bg[2];
#pragma unroll
        for (int ch = 0; ch < GS2M_NUM_FEATURES; ch++) {
            const int cch = 3 + ch;
            out_buffer[ch * HW + pix] = (ch < FC && ch < fc) ? acc[(cch >> 1) < NP ? (cch >> 1) : 0][cch & 1] : 0.0f;
        }
    }
}

}  // namespace

void gs2m_launch_blend_fwd_q(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                             const BinningState& b, const ImageState& im, float* out_color, float* out_buffer,
                             int* out_observe, hipStream_t s) {
    const int tiles = tiles_x * tiles_y;
    const int grid = ((tiles + 7) / 8) * 32;
    const int fct = fc <= 1 ? 1 : (fc <= 5 ? 5 : (fc <= 9 ? 9 : 10));
#define GS2M_FWDQ(FC)                                                                                                   \
    blend_fwd_q_kernel<FC><<<grid, 64, 0, s>>>(im.ranges, b.qlist, im.qcount, g.rec, W, H, tiles_x, tiles, bg, fc, out_color, \
                                               out_buffer, im.final_T, im.n_contrib, out_observe, im.qlast)
    switch (fct) {
        case 1: GS2M_FWDQ(1); break;
        case 5: GS2M_FWDQ(5); break;
        case 9: GS2M_FWDQ(9); break;
        default: GS2M_FWDQ(10); break;
    }
#undef GS2M_FWDQ
}
