// Internal declarations shared by the HIP translation units of libgs2m_raster.so.
// gfx950 (MI355X) only: wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gs2m_raster.h"

// Non-temporal ("nt") global accesses for the big read-once / write-once streams: the SH coefficients on their way into the
// preprocess kernel (192 B per Gaussian), the gradient rows on their way into the per-Gaussian backward, dL/dSH on its way out.
// Without the hint such a stream pushes the lines other kernels come back for (the blend records the preprocess kernel is
// writing, which the emit kernel reads next) out of the caches: preprocess 88 -> 79 us and emit 42 -> 39 us for the SH loads alone,
// per-Gaussian backward 160 -> 156 us (same-box A/B, NOTEBOOK.md round 6).  Measured WITHOUT effect or worse, and left as plain
// accesses: the small per-Gaussian inputs, record / sh_dir / image stores, list and pixel-gradient loads of the blend kernels
// (ALU-bound), key loads of the radix sort, slot loads of the tile sort; nt STORES of the gradient rows make their reader slower.
typedef float gs2m_v4f __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ T gs2m_ldnt(const T* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float4 gs2m_ldnt(const float4* p) {
    const gs2m_v4f t = __builtin_nontemporal_load(reinterpret_cast<const gs2m_v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
template <typename T> __device__ __forceinline__ void gs2m_stnt(T* p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void gs2m_stnt(float4* p, float4 v) {
    const gs2m_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<gs2m_v4f*>(p));
}

#define GS2M_ALIGN 256
#define GS2M_WAVE 64

// ---- blend record: 8 x float4 = 128 B per Gaussian (one aligned HBM line) -------------
// q0: x, y, A, B          pixel-space mean, conic (A = conic.x, B = conic.y)
// q1: C, opacity, hx, hy  conic.z, opacity, half extents of the alpha >= 1/255 ellipse (the extents: diagnostic only since round 5)
// q2: 0, rmin, rwh, t2    (unused), tile rect min (x | y << 16), rect (w | h << 16) -- also in GeomState::rect --,
//                        t2 = upper bound of A dx^2 + 2B dx dy + C dy^2 where alpha can reach 1/255
// q3..q6: the 13 blended channels, contiguous: r, g, b (SH-evaluated or precomputed), features[0..9], 3 pad
//         floats -- channel c is float 12 + c of the record, so a kernel blending fc features stages
//         3 + ceil((3 + fc) / 4) quads and can feed whole quads to the matrix pipe
// q7: unused
#define REC_Q 8
#define REC_GEO0 0
#define REC_GEO1 1
#define REC_BIN 2
#define REC_CH 3
#define REC_AUX 7
// the binned values carry the instance's quadrant-hit mask above the Gaussian id
#define GS2M_GID_BITS 28
#define GS2M_GID_MASK 0x0FFFFFFFu
// HEAVY Gaussians: at least this many tile instances.  A wave of the emit kernel and of the per-Gaussian backward owns 64 consecutive
// Gaussians; trained models keep their big splats together in index order (the initial points, whole densification generations), so
// one wave would expand thousands of instances and add up tens of thousands of gradient rows while the rest of the chip waits.
// Heavy Gaussians are taken out of those waves: their instances are cut into UNITS of 64 (a Gaussian owns ceil(instances / 64) whole
// units, numbered in index order through the block scan), a wave per unit expands them (emit_heavy_kernel), every instance of a unit
// has 4 gradient rows reserved (rows 256 u .. 256 u + 255: the heavy rows come first, the waves' dense rows follow), a wave per unit
// adds the rows up in the backward (heavy_reduce_kernel) and the Gaussian's thread adds its units' sums.  GeomState::gauss_rows of a
// heavy Gaussian = GS2M_ROWS_BIG | its first unit.  (40: above every Gaussian of the bench clouds, whose largest covers 36 tiles.)
// In a CROWDED wave -- the 64 Gaussians hold more than GS2M_CROWDED_WAVE instances between them, heavy ones not counted: a run of
// medium-sized splats -- the bar drops to GS2M_HEAVY_TILES_CROWDED, so that what a wave keeps for itself stays below ~450 instances
// (7 steps of the emit loop, ~28 windows of rows in the backward; the waves of the bench clouds hold 173 on average, 296 at most:
// tools/wave_sums.py; on the C4 view 48 / 512 -> 40 / 320 took the per-Gaussian backward from 110 to 88 us).  gs2m_heavy() below: the preprocess and the emit kernel decide
// alike (same waves), the backward reads the decision off gauss_rows.
#define GS2M_HEAVY_TILES 40u
#define GS2M_HEAVY_TILES_CROWDED 8u
#define GS2M_CROWDED_WAVE 320u
#define GS2M_ROWS_BIG 0x80000000u
#define GS2M_UNIT 64u
struct HeavyUnit {      // 80 bytes per unit, at the end of the binning buffer
    uint32_t gid;       // the Gaussian the unit belongs to
    uint32_t off;       // first emission slot of that Gaussian
    uint32_t pad[2];
    uint8_t pop[GS2M_UNIT];  // gradient rows (set quadrant bits) of each instance of the unit: rows 256 u + 4 k .. + pop[k] - 1 are written
};

// per (instance, quadrant) partial-gradient row produced by the blend backward (floats):
// 0 mx, 1 my, 2 |mx|, 3 |my|, 4 cxx, 5 cxy, 6 cyy, 7 dopacity, 8..10 dcolor, 11.. dfeature
#define ROW_GEOM 8
#define ROW_COL 8
#define ROW_FEAT 11

static inline size_t gs2m_align_up(size_t x, size_t a = GS2M_ALIGN) { return (x + a - 1) & ~(a - 1); }

// Round 5: no global sort.  Gaussians are binned in INDEX order (count -> scan -> fill, binning.hip) and every tile's span is
// sorted by (depth, id) on chip (tile_sort.hip); rows of a Gaussian's gradient partials are numbered in index order too.
struct GeomState {
    float4* rec;             // P * 8
    uint32_t* tiles_touched; // P (by Gaussian id): tiles emitted for the Gaussian (rect w x h)
    uint2* rect;             // P: {tile rect min x | y << 16, w | h << 16} of the emitted rectangle (0, 0 when nothing is emitted)
    uint32_t* depth_key;     // P: fp32 bits of the view depth, 0xFFFFFFFF when culled
    uint8_t* clamped;        // P
    float* sh_dir;           // P * 9: d(SH colour)/d(view direction) of a visible Gaussian, {dRdx, dRdy, dRdz}[rgb] (preprocess -> gaussian_bwd)
    uint32_t* counters;      // 64 u32 (GS2M_CNT_* below; counters[0] = num_rendered as emit_kernel's offsets add up: debug mode compares)
    uint32_t* gauss_rows;    // P: gradient rows of each Gaussian, or GS2M_ROWS_BIG | first unit of a heavy one (emit_kernel)
    uint32_t* block_tt;      // ceil(P / 256): tiles_touched summed over blocks of 256 Gaussians (preprocess kernel)
    uint32_t* block_pref;    // ceil(P / 256): exclusive prefix of block_tt (scan kernel): first emission offset of a block
    uint32_t* block_hu;      // ceil(P / 256): heavy units of the block's Gaussians (preprocess kernel)
    uint32_t* block_hupref;  // ceil(P / 256): exclusive prefix of block_hu (scan kernel)
    uint32_t* wave_rows;     // ceil(P / 64): gradient rows of each wave of 64 consecutive Gaussians, heavy ones not counted (emit_kernel)
    uint32_t* wave_rowbase;  // ceil(P / 64): exclusive prefix of wave_rows (rowscan_kernel): first gradient row of the wave
    uint32_t* tile_hist;     // GS2M_HIST_COPIES x 1024: the tile sort's digit histograms, counted by emit_kernel (zeroed by the preprocess kernel)
    size_t total_bytes;      // including alignment slack
};
struct BinningState {
    uint32_t* keys_unsorted; // R: tile id of the instance at each emission slot (index order)
    uint4* e_rec;            // R: per emission slot {Gaussian id | quadrant-hit mask << 28, first gradient row relative to the emit wave's,
                             //    depth key, 0}: ONE 16-byte gather per instance in the per-tile sort
    uint32_t* sort_keyA;     // R (radix sort ping buffer)
    uint32_t* sort_valA;     // R
    uint32_t* tile_keys;     // R: sorted tile ids
    uint32_t* slot_sorted;   // R: emission slots in (tile, index) order -- the stable tile sort's values
    uint32_t* point_list;    // R: Gaussian id | mask << 28 sorted by (tile, depth, id) -- the reference's point_list (+ mask bits)
    uint2* qlist;            // 4R: per (tile, 8x8 quadrant) compacted lists {Gaussian id | mask << 28, position in the tile list};
                             //     the list of (tile, q) starts at 4 * ranges[tile].x + q * (tile list length)
    uint32_t* qrow;          // 4R: gradient row of each list entry (parallel to qlist)
    char* temp;              // radix sort scratch
    size_t temp_bytes;
    HeavyUnit* hrec;         // U heavy units (behind everything else: no other offset depends on U)
    size_t total_bytes;
};
struct ImageState {
    float* final_T;      // N
    uint32_t* n_contrib; // N
    uint2* ranges;       // tiles (written by the tile-sort kernel from ranges_raw)
    uint32_t* ranges_raw; // tiles * 2: per tile {~first position, last position + 1} as atomicMax targets of the tile sort's last pass; 0, 0 = untouched
    uint32_t* qcount;    // tiles * 4: entries in each quadrant list
    uint32_t* qlast;     // tiles * 4: entries up to and including the quadrant's last contributor (forward -> backward)
    size_t total_bytes;
};

// Up to three word ranges a kernel zeroes on the side (grid-stride over all its threads): saves the separate zero-fill
// launches in front of the kernels that consume them.
struct ZeroJobs {
    uint32_t* p[3];
    size_t words[3];
};
#ifdef __HIPCC__
__device__ __forceinline__ void gs2m_zero_jobs(const ZeroJobs& z, size_t thread, size_t threads) {
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (size_t i = thread; i < z.words[k]; i += threads) z.p[k][i] = 0u;
}
#endif

#ifdef __HIPCC__
// ---- SH rows of a 256-Gaussian block -> LDS (row stride 49 floats: conflict-free column reads) --------------------
// (the way back, dL/dSH, leaves as basis x gradient: gaussian_bwd.hip, sh_outer_store)
// SH coefficients are 192 B per Gaussian (M = 16): a thread-per-Gaussian access has a 192-B lane stride, so the block
// streams its rows with coalesced float4 accesses and every thread then works on its own LDS row.  Two source
// layouts: one (P,16,3) tensor, or -- as the reference model stores its parameters (scene/gaussian_model.py:
// _features_dc (P,1,3), _features_rest (P,15,3)) -- the DC and the rest separately (rest != nullptr), which saves
// the caller a 192-B-per-Gaussian concatenation per view and its backward.
__device__ __forceinline__ void gs2m_stage_sh(const float* __restrict__ shs, const float* __restrict__ rest, int P,
                                              float* __restrict__ s_sh) {
    const int tid = threadIdx.x;
    if (rest == nullptr) {
        const size_t base4 = (size_t)blockIdx.x * 256 * 12, lim4 = (size_t)P * 12;  // float4 index of the block's first row
        const float4* g4 = reinterpret_cast<const float4*>(shs);
        float4 t[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const size_t k = base4 + tid + 256 * i;
            t[i] = k < lim4 ? gs2m_ldnt(g4 + k) : make_float4(0.f, 0.f, 0.f, 0.f);  // 192 bytes per Gaussian, read once (see gs2m_ldnt)
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int e = 4 * (tid + 256 * i);
            const int row = e / 48, col = e - row * 48;
            float* d = s_sh + row * 49 + col;
            d[0] = t[i].x; d[1] = t[i].y; d[2] = t[i].z; d[3] = t[i].w;
        }
    } else {
        // DC: 3 floats per row, 768 per block, scalar loads; rest: 45 floats per row, 2880 float4 per block
        const size_t dbase = (size_t)blockIdx.x * 768, dlim = (size_t)P * 3;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int e = tid + 256 * i;
            const float v = dbase + e < dlim ? shs[dbase + e] : 0.f;
            s_sh[(e / 3) * 49 + (e % 3)] = v;
        }
        // rest: float4 k4 = tid + 256 i of the block's 2880; its first element e = 4 k4 sits in row e / 45 at column e % 45.
        // One division per thread: a step of 256 float4 is 1024 elements = 22 rows + 34 columns.  A block that lies
        // completely inside the tensor (all but the last) loads without per-element bounds tests (wave-uniform branch).
        const size_t rbase = (size_t)blockIdx.x * 11520, rlim = (size_t)P * 45;
        const bool full = rbase + 11520 <= rlim;
        float4 t[12];
        if (full) {
            const float4* g4 = reinterpret_cast<const float4*>(rest + rbase);
#pragma unroll
            for (int i = 0; i < 11; i++) t[i] = gs2m_ldnt(g4 + tid + 256 * i);
            t[11] = tid < 64 ? gs2m_ldnt(g4 + tid + 2816) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const int k4 = tid + 256 * i;
                const size_t ge = rbase + 4 * (size_t)k4;
                t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k4 < 2880) {
                    if (ge + 3 < rlim) {
                        t[i] = *reinterpret_cast<const float4*>(rest + ge);
                    } else {  // the tensor ends inside this float4
                        if (ge < rlim) t[i].x = rest[ge];
                        if (ge + 1 < rlim) t[i].y = rest[ge + 1];
                        if (ge + 2 < rlim) t[i].z = rest[ge + 2];
                    }
                }
            }
        }
        int row = (4 * tid) / 45, col = 4 * tid - 45 * row;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            if (i < 11 || tid < 64) {
                const float v[4] = {t[i].x, t[i].y, t[i].z, t[i].w};
                int a = row * 49 + 3 + col, left = 45 - col;  // elements left in this row
#pragma unroll
                for (int c = 0; c < 4; c++) s_sh[a + c + (c >= left ? 4 : 0)] = v[c];
            }
            row += 22; col += 34;
            if (col >= 45) { col -= 45; row += 1; }
        }
    }
}
#endif

// carve typed arrays out of one byte buffer (base may be unaligned; pass nullptr to size)
GeomState gs2m_carve_geom(char* base, size_t P);
BinningState gs2m_carve_binning(char* base, size_t R, size_t temp_bytes, size_t heavy_units);
size_t gs2m_binning_temp_bytes(size_t R, int tile_bits);
ImageState gs2m_carve_image(char* base, size_t N, size_t tiles);

// hand-written onesweep radix sort (radix_sort.hip).  Rounds 1-4 sorted the Gaussians by depth and the instances by tile with
// it; since round 5 the rasterizer buckets by tile and sorts each tile on chip (binning.hip, tile_sort.hip), and the only
// user left is distCUDA2's Morton order (knn.hip).
size_t gs2m_radix_temp_bytes(size_t n, int total_bits);
struct SideBuckets {
    const uint32_t* tt;
    uint32_t* buckets;
    uint32_t* supers;
};
struct SideSum {
    const uint32_t* tt;
    uint32_t* acc;
    uint32_t* landing;
};
hipError_t gs2m_radix_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* kin, const uint32_t* vin, uint32_t* kA,
                                 uint32_t* vA, uint32_t* kB, uint32_t* vB, size_t n, int total_bits, bool prezeroed, hipStream_t s,
                                 SideSum sum = SideSum{nullptr, nullptr, nullptr}, uint32_t* range_raw = nullptr,
                                 const uint32_t* ext_hist = nullptr, SideBuckets sb = SideBuckets{nullptr, nullptr, nullptr});
#define GS2M_HIST_COPIES 8
#define GS2M_HIST_COPY_WORDS 1024
void gs2m_radix_plan(int total_bits, int* npass, int bits[4], int shift[4]);
void gs2m_radix_zero_region(void* temp, size_t n, int total_bits, uint32_t** ptr, size_t* words);

// words the host reads back through a mapped pinned block (api.hip): [0] num_rendered and [1] heavy units (ONE 8-byte store: the
// second is there when the first is), [2] prefiltered violation flag, [3] gradient rows + 1 (0 = not landed yet)
#define GS2M_LAND_R 0
#define GS2M_LAND_HUNITS 1
#define GS2M_LAND_PREFILTERED 2
#define GS2M_LAND_ROWS 3
// GeomState::counters: [0] num_rendered as the emit kernel's offsets add up (debug mode), [1] num_rendered, [2] gradient rows, [3] heavy units
#define GS2M_CNT_ROWS 2
#define GS2M_CNT_HUNITS 3
// a tile span of 513 .. 1024 entries exists / one beyond 1024 (set by the first kernel of tile_sort.hip with plain stores of 1, zeroed by
// blockscan_kernel): the kernels for those spans, launched over all tiles whatever the frame, leave at once when there is none
#define GS2M_CNT_SPAN_MID 4
#define GS2M_CNT_SPAN_LONG 5

// kernel launchers
void gs2m_launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                            const float* rotations, const float* opacities, const float* shs, const float* shs_rest,
                            const float* cov3D_precomp, const float* colors_precomp, const float* features,
                            const float* viewmatrix, const float* projmatrix, const float* cam_pos, int W, int H,
                            float tan_fovx, float tan_fovy, float focal_x, float focal_y, int tiles_x, int tiles_y,
                            int* radii, int* observe_zero, const GeomState& g, int shrink, const ZeroJobs& zero, hipStream_t s);
// binning.hip: blockscan (publishes num_rendered; block prefixes of tiles_touched), emit (instances in index order: tile keys,
// quadrant masks, gradient-row numbering, the tile sort's digit counts) + rowscan (first gradient row of every wave)
void gs2m_launch_blockscan(int P, const GeomState& g, uint32_t* landing, hipStream_t s);
void gs2m_launch_emit(int P, int W, int H, int tiles_x, int tile_bits, const GeomState& g, const BinningState& b, uint32_t heavy_units,
                      uint32_t crowded, uint32_t* landing, const ZeroJobs& zero, hipStream_t s);
// the blocks' heavy-unit counts once more with the crowded-wave rule off (the preprocess kernel counted them with it on)
void gs2m_launch_recount_heavy(int P, const GeomState& g, hipStream_t s);
// gaussian_bwd.hip: the rows of every heavy unit added up into the unit's first row (before gaussian_bwd_kernel); heavy_units < 0:
// not known on the host (a fixed grid reads the count on the device)
void gs2m_launch_heavy_reduce(float* rows, int rowf, const BinningState& b, const GeomState& g, long long heavy_units, hipStream_t s);
// tile_sort.hip: every tile's span (stable radix sort by tile: index order) sorted by (depth, id) on chip, then split into the four
// quadrant lists; writes ranges[] from ranges_raw
void gs2m_set_tile_sort_policy_impl(int policy);
void gs2m_launch_tile_sort(size_t tiles, int tiles_x, int tiles_y, size_t R, const BinningState& b, const ImageState& im, const GeomState& g, hipStream_t s);
hipError_t gs2m_zero_async(void* p, size_t bytes, hipStream_t s);

void gs2m_launch_blend_fwd_q(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                             const BinningState& b, const ImageState& im, float* out_color, float* out_buffer,
                             int* out_observe, hipStream_t s);
void gs2m_launch_blend_bwd_q(int W, int H, int tiles_x, int tiles_y, int fc, const float* bg, const GeomState& g,
                             const BinningState& b, const ImageState& im, const float* grad_color,
                             const float* grad_buffer, float* rows, hipStream_t s);
int gs2m_row_floats(int fc);  // floats per partial-gradient row: 11 + fc, padded to a multiple of 4
void gs2m_launch_gaussian_bwd(int P, int D, int M, const float* means3D, const float* shs, const float* shs_rest,
                              const float* colors_precomp,
                              const float* scales, float scale_modifier, const float* rotations,
                              const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                              const float* campos, int W, int H, float tan_fovx, float tan_fovy, const int* radii,
                              int fc, const GeomState& g, const float* rows, int rowf, bool have_rows,
                              float* dL_dmeans2D, float* dL_dconics, float* dL_dopacities, float* dL_dcolors,
                              float* dL_dmeans3D, float* dL_dcov3D, float* dL_dshs, float* dL_dshs_rest, float* dL_dscales,
                              float* dL_drots, float* dL_dfeatures, hipStream_t s);
void gs2m_launch_prefiltered_check(int P, const float* means3D, const float* viewmatrix, uint32_t* flag, hipStream_t s);
void gs2m_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

#ifdef __HIPCC__
// ---- device helpers -----------------------------------------------------------------
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// DPP cross-lane move (gfx9 encodings).  Lanes masked off by row_mask/bank_mask get 0.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_mov0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
#define DPP_QUAD_XOR1 0xB1     // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E     // quad_perm [2,3,0,1]
#define DPP_QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_ROW_SHR(n) (0x110 + (n))

// Sum over the 64 lanes of a wave; the total is valid in lanes 48..63 (row 3).
__device__ __forceinline__ float wave_sum_row3(float v) {
    v += dpp_mov0<DPP_QUAD_XOR1>(v);
    v += dpp_mov0<DPP_QUAD_XOR2>(v);
    v += dpp_mov0<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov0<DPP_ROW_MIRROR>(v);
    v += dpp_mov0<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_mov0<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

// Workgroup barrier that is safe at loop headers.  hipcc (ROCm 7.2, gfx950) emitted a bare
// `s_barrier` for a __syncthreads() at the top of a loop whose back edge ends in LDS stores: the
// release half (s_waitcnt lgkmcnt(0)) was missing, so other waves could pass the barrier and read
// LDS before this wave's ds_writes had landed (seen as run-to-run differences in ~0.04 % of the
// gradient rows at 1M Gaussians; tests/test_fullsize_gpu.py).  The inline-asm wait is invisible to
// the waitcnt pass and therefore always kept.
__device__ __forceinline__ void gs2m_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ int gs2m_sync_count(bool pred) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return __syncthreads_count(pred);
}
__device__ __forceinline__ int gs2m_sync_or(bool pred) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return __syncthreads_or(pred);
}

// Is the Gaussian of this lane heavy?  `cnt`: its tile instances; called by all 64 lanes of a wave of 64 consecutive Gaussians.
// `crowded`: GS2M_CROWDED_WAVE, or GS2M_CROWDED_OFF when the forward found that the crowded-wave rule would reserve more rows than
// the frame can justify (api.hip: a frame whose waves are ALL crowded has no imbalance to repair).
#define GS2M_CROWDED_OFF 0xFFFFFFFFu
__device__ __forceinline__ bool gs2m_heavy(uint32_t cnt, uint32_t crowded) {
    uint32_t light = cnt < GS2M_HEAVY_TILES ? cnt : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) light += (uint32_t)__shfl_xor((int)light, d, 64);
    return cnt >= (light > crowded ? GS2M_HEAVY_TILES_CROWDED : GS2M_HEAVY_TILES) && cnt < (1u << 29);
}

// Inclusive prefix sum over the 64 lanes of a wave (u32).
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// alpha evaluation shared bit-for-bit by the forward and backward blend kernels:
//   power = -0.5f * (A*dx*dx + C*dy*dy) - B*dx*dy      (CR/forward.cu:326-329)
// evaluated UNFUSED and in the reference's written order.  For splats hundreds of pixels long
// the three terms are O(1e2..1e3) and cancel to O(1); keeping the written order makes the
// rounding of `power` identical to the CPU oracle's, so parity does not degrade with the
// conditioning of the conic (only exp() differs, by ~1 ulp).  Costs 4 VALU ops per evaluated
// pixel x Gaussian pair over a pre-scaled FMA form.
#define GS2M_LOG2E 1.4426950408889634f
__device__ __forceinline__ float gs2m_power(float dx, float dy, float A, float B, float C) {
    const float t1 = (A * dx) * dx;
    const float t2 = (C * dy) * dy;
    const float t3 = (B * dx) * dy;
    return (-0.5f * (t1 + t2)) - t3;
}
// Does the region {q(dx, dy) = A dx^2 + 2B dx dy + C dy^2 <= t2} around (gx, gy) reach the pixel rectangle
// [x0, x1] x [y0, y1]?  Exact up to the margins t2 carries (preprocess.hip): the centre lies inside, or the
// minimum of q over the four edges is within t2.  Instances that fail cannot contribute to any pixel of the
// rectangle; t2 >= 3e38 marks Gaussians whose culling is disabled (indefinite / ill-conditioned conic).
// The blend loops are issue bound and this test runs once per (instance, quadrant), so it is written for
// instruction count: v_rcp_f32 (1 ulp, second-order effect on q at the clamped minimiser), v_med3 clamps.
//
// Minimum of q over the two edges dx = l, dx = u with the free coordinate clamped to [fl, fu]
// (a, c = the conic entries of the fixed and the free coordinate, b2 = 2B).
__device__ __forceinline__ float gs2m_edge_pair_qmin(float a, float b2, float c, float l, float u, float fl, float fu) {
    const float nhr = -0.5f * __builtin_amdgcn_rcpf(c);
    const float bl = b2 * l, bu = b2 * u;
    const float fml = __builtin_amdgcn_fmed3f(bl * nhr, fl, fu), fmu = __builtin_amdgcn_fmed3f(bu * nhr, fl, fu);
    const float ql = __builtin_fmaf(__builtin_fmaf(c, fml, bl), fml, (a * l) * l);
    const float qu = __builtin_fmaf(__builtin_fmaf(c, fmu, bu), fmu, (a * u) * u);
    return fminf(ql, qu);
}
__device__ __forceinline__ bool gs2m_reaches_rect(float gx, float gy, float A, float B, float C, float t2, float x0,
                                                  float x1, float y0, float y1) {
    const float lx = x0 - gx, ux = x1 - gx, ly = y0 - gy, uy = y1 - gy;  // rectangle relative to the centre
    const bool inside = lx <= 0.f && ux >= 0.f && ly <= 0.f && uy >= 0.f;
    const float b2 = B + B;
    const float qmin = fminf(gs2m_edge_pair_qmin(A, b2, C, lx, ux, ly, uy), gs2m_edge_pair_qmin(C, b2, A, ly, uy, lx, ux));
    return !(t2 < 3.0e38f) || inside || qmin <= __builtin_fmaf(1.0e-3f, fabsf(qmin), t2);
}
// gs2m_reaches_rect for the four 8x8 quadrants of the 16x16 tile whose first pixel is (x0, y0), sharing what the four
// tests have in common (two reciprocals instead of eight, the per-line products): bit q of the result is EXACTLY
// gs2m_reaches_rect(..., x0 + 8 (q & 1), + 7, y0 + 8 (q >> 1), + 7) -- same operations on the same values per edge.
__device__ __forceinline__ uint32_t gs2m_reaches_quads(float gx, float gy, float A, float B, float C, float t2, float x0, float y0) {
    const float b2 = B + B;
    const float nhrC = -0.5f * __builtin_amdgcn_rcpf(C), nhrA = -0.5f * __builtin_amdgcn_rcpf(A);
    float xl[4], yl[4], blx[4], bly[4], basex[4], basey[4], mx[4], my[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float o = (float)((i >> 1) * 8 + (i & 1) * 7);  // 0, 7, 8, 15
        xl[i] = (x0 + o) - gx; yl[i] = (y0 + o) - gy;
        blx[i] = b2 * xl[i]; bly[i] = b2 * yl[i];
        basex[i] = (A * xl[i]) * xl[i]; basey[i] = (C * yl[i]) * yl[i];
        mx[i] = blx[i] * nhrC;  // unclamped minimiser in y along the vertical line dx = xl[i]
        my[i] = bly[i] * nhrA;  // ... in x along the horizontal line dy = yl[i]
    }
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int qx = q & 1, qy = q >> 1;
        const float lx = xl[2 * qx], ux = xl[2 * qx + 1], ly = yl[2 * qy], uy = yl[2 * qy + 1];
        const bool inside = lx <= 0.f && ux >= 0.f && ly <= 0.f && uy >= 0.f;
        float qe[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {  // the two vertical edges over [ly, uy]
            const int i = 2 * qx + e;
            const float fm = __builtin_amdgcn_fmed3f(mx[i], ly, uy);
            qe[e] = __builtin_fmaf(__builtin_fmaf(C, fm, blx[i]), fm, basex[i]);
        }
        const float qv = fminf(qe[0], qe[1]);
#pragma unroll
        for (int e = 0; e < 2; e++) {  // the two horizontal edges over [lx, ux]
            const int k = 2 * qy + e;
            const float fm = __builtin_amdgcn_fmed3f(my[k], lx, ux);
            qe[e] = __builtin_fmaf(__builtin_fmaf(A, fm, bly[k]), fm, basey[k]);
        }
        const float qmin = fminf(qv, fminf(qe[0], qe[1]));
        const bool hit = !(t2 < 3.0e38f) || inside || qmin <= __builtin_fmaf(1.0e-3f, fabsf(qmin), t2);
        mask |= hit ? (1u << q) : 0u;
    }
    return mask;
}
// The same test with the work of one instance split over two lanes (lane and lane ^ 32): `swap` lanes take
// the horizontal edges by exchanging the roles of x and y, then the halves are combined.
__device__ __forceinline__ bool gs2m_reaches_rect_split(bool swap, float gx, float gy, float A, float B, float C,
                                                        float t2, float x0, float x1, float y0, float y1) {
    const float lx = x0 - gx, ux = x1 - gx, ly = y0 - gy, uy = y1 - gy;
    const bool inside = lx <= 0.f && ux >= 0.f && ly <= 0.f && uy >= 0.f;
    const float q = gs2m_edge_pair_qmin(swap ? C : A, B + B, swap ? A : C, swap ? ly : lx, swap ? uy : ux,
                                        swap ? lx : ly, swap ? ux : uy);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(q), __float_as_uint(q), false, false);
    const float qmin = fminf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // own half and the partner's
    return !(t2 < 3.0e38f) || inside || qmin <= __builtin_fmaf(1.0e-3f, fabsf(qmin), t2);
}

// exp(x) for x <= 0 through v_exp_f32
__device__ __forceinline__ float gs2m_exp(float x) { return __builtin_amdgcn_exp2f(x * GS2M_LOG2E); }
#endif
