// Per-tile front-to-back alpha compositing (forward) for gfx950.
//
// Semantics: renderCUDA, diff-gaussian-rasterization/cuda_rasterizer/forward.cu:246-372.
// One 256-thread workgroup per 16x16 tile as in the reference, but organised for 64-lane
// wavefronts: each of the 4 waves owns one 8x8 pixel quadrant.  Per batch of 256 sorted
// instances the workgroup gathers the 128-B blend records cooperatively (8 lanes x 16 B per
// record, full-line reads) into LDS -- geometry AND colour/feature rows, the reference only
// stages geometry and re-reads colours from global per pixel.  Each wave then tests 64
// instances at a time (one per lane) against its quadrant's rectangle, takes a 64-bit
// ballot, and walks only the set bits; the per-instance data is then read from LDS with a
// wave-uniform address (broadcast).  Skipped instances cannot contribute (alpha < 1/255
// over the whole quadrant, see preprocess.hip), so results are unchanged.
// The colour/feature accumulation runs as packed fp32 (v_pk_fma_f32: two channels per instruction, the
// weight broadcast) -- measured on gfx950 a packed FMA issues at the rate of a scalar one, and the matrix
// pipe is no alternative here: MFMA and VALU instructions do not overlap on a SIMD (tools/micro/).
// `observe` (pixels an instance contributes to with T > 0.5) is one ballot popcount per survivor, written into a
// per-lane register with v_writelane, added to the instance's LDS counter once per lane and 64 instances, and stored
// once per instance in emission order -- no global atomics (forward.cu:348-350 uses one atomicAdd per pixel);
// binning.hip:observe_kernel reduces the per-instance counts per Gaussian.
#include "common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int BATCH = 128;

template <int FC>  // feature channels blended (compile time); runtime fc <= FC
__global__ void __launch_bounds__(256) blend_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, int W,
    int H, int tiles_x, const float* __restrict__ bg, int fc, float* __restrict__ out_color,
    float* __restrict__ out_buffer, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    uint32_t* __restrict__ inst_obs) {
    constexpr int NC = 3 + FC;            // blended channels: r, g, b, features
    constexpr int KQ = (NC + 3) / 4;      // channel quads of the record
    constexpr int NP = (NC + 1) / 2;      // channel pairs accumulated
    constexpr int NQ = 3 + KQ;            // record quads staged: geo0, geo1, bin, channels
    __shared__ float4 s_v[NQ][BATCH];
    __shared__ uint32_t s_gid[BATCH];
    __shared__ uint32_t s_slot[BATCH];
    __shared__ int s_obs[BATCH];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.x;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int px = tile_x * GS2M_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tile_y * GS2M_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)(tile_x * GS2M_TILE + (wave & 1) * 8), bx1 = bx0 + 7.0f;
    const float by0 = (float)(tile_y * GS2M_TILE + (wave >> 1) * 8), by1 = by0 + 7.0f;

    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);

    float T = 1.0f;
    uint32_t last_contributor = 0;
    v2f acc[NP];  // acc[k] = channels 2k, 2k + 1
#pragma unroll
    for (int k = 0; k < NP; k++) acc[k] = v2f{0.f, 0.f};
    bool done = !inside;
    int prev_cnt = 0;

    for (int base = 0; base < len; base += BATCH) {
        // (S1) everyone finished the previous batch; vote on early exit (forward.cu:300-302)
        const int num_done = gs2m_sync_count(done);
        if (prev_cnt > 0 && tid < prev_cnt) {
            const int o = s_obs[tid];
            if (o != 0) inst_obs[s_slot[tid]] = (uint32_t)o;
        }
        prev_cnt = 0;
        if (num_done == 256) break;
        const int cnt = min(BATCH, len - base);
        if (tid < cnt) s_gid[tid] = point_list[range.x + base + tid];
        gs2m_sync();  // (S2) s_obs/s_slot of the previous batch consumed, gids visible
        {
            const int q = tid & 7;
            if (q < NQ) {
#pragma unroll
                for (int r = 0; r < BATCH / 32; r++) {
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
            if (tid < BATCH) s_obs[tid] = 0;
        }
        gs2m_sync();  // (S3) batch staged
        prev_cnt = cnt;

        if (__builtin_amdgcn_ballot_w64(!done) != 0ull) {
            for (int sub = 0; sub < cnt; sub += GS2M_WAVE) {
                const int j = sub + lane;
                bool hit = false;
                if (j < cnt) {
                    const float4 a = s_v[REC_GEO0][j], b = s_v[REC_GEO1][j];
                    hit = gs2m_reaches_rect(a.x, a.y, a.z, a.w, b.x, s_v[REC_BIN][j].w, bx0, bx1, by0, by1);
                }
                unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                int myobs = 0;  // observe count of instance sub + lane in this quadrant
                while (mask) {
                    const int bit = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int jj = sub + bit;  // wave-uniform
                    const float4 a = s_v[REC_GEO0][jj], b = s_v[REC_GEO1][jj];
                    float4 c[KQ];  // requested before the evaluation: their LDS latency hides behind it
#pragma unroll
                    for (int q = 0; q < KQ; q++) c[q] = s_v[REC_CH + q][jj];
                    const float dx = a.x - pxf, dy = a.y - pyf;
                    const float p2 = gs2m_power(dx, dy, a.z, a.w, b.x);
                    const float alpha = fminf(0.99f, b.y * gs2m_exp(p2));
                    bool contrib = !done && (p2 <= 0.0f) && (alpha >= 1.0f / 255.0f);
                    const float test_T = T * (1.0f - alpha);
                    if (contrib && test_T < 0.0001f) {
                        done = true;
                        contrib = false;
                    }
                    const float tw = contrib ? T : 0.f;  // branch-free: non-contributing lanes add 0
                    const float w = alpha * tw;
                    const v2f ww = {w, w};
#pragma unroll
                    for (int q = 0; q < KQ; q++) {
                        if (4 * q < NC) acc[2 * q] = __builtin_elementwise_fma(v2f{c[q].x, c[q].y}, ww, acc[2 * q]);
                        if (4 * q + 2 < NC) acc[2 * q + 1] = __builtin_elementwise_fma(v2f{c[q].z, c[q].w}, ww, acc[2 * q + 1]);
                    }
                    if (contrib) last_contributor = (uint32_t)(base + jj + 1);
                    const int seen = (int)__popcll(__builtin_amdgcn_ballot_w64(tw > 0.5f));  // wave-uniform
                    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(myobs) : "s"(seen), "s"(bit) : "m0");  // myobs[lane == bit] = seen
                    if (contrib) T = test_T;
                }
                if (myobs != 0) atomicAdd(&s_obs[j], myobs);  // one LDS atomic per lane and 64 instances, 4 waves
                // every pixel of the quadrant finished: skip the rest of the batch (the reference votes once per
                // 256-instance batch, forward.cu:300-302; finished pixels ignore later instances either way)
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            }
        }
    }
    gs2m_sync();
    if (prev_cnt > 0 && tid < prev_cnt) {
        const int o = s_obs[tid];
        if (o != 0) inst_obs[s_slot[tid]] = (uint32_t)o;
    }

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
            const int c = 3 + ch;
            out_buffer[ch * HW + pix] = (ch < FC && ch < fc) ? acc[(c >> 1) < NP ? (c >> 1) : 0][c & 1] : 0.0f;
        }
    }
}

}  // namespace

void gs2m_launch_blend_fwd(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                           const BinningState& b, const ImageState& im, float* out_color, float* out_buffer,
                           hipStream_t s) {
    const int tiles = tiles_x * tiles_y;
    const int fct = fc <= 1 ? 1 : (fc <= 5 ? 5 : (fc <= 9 ? 9 : 10));
#define GS2M_FWD(FC)                                                                                              \
    blend_fwd_kernel<FC><<<tiles, 256, 0, s>>>(im.ranges, b.point_list, g.rec, W, H, tiles_x, bg, fc, out_color, \
                                               out_buffer, im.final_T, im.n_contrib, b.inst_obs)
    switch (fct) {
        case 1: GS2M_FWD(1); break;
        case 5: GS2M_FWD(5); break;
        case 9: GS2M_FWD(9); break;
        default: GS2M_FWD(10); break;
    }
#undef GS2M_FWD
}
