// extern "C" entry points of libgs2m_raster.so (see include/gs2m_raster.h for the contract
// and the reference interfaces each one replaces).  Orchestration mirrors
// CudaRasterizer::Rasterizer::forward/backward (cuda_rasterizer/rasterizer_impl.cu:185-438)
// with the pipeline described in binning.hip.
#include "common.h"
#include <atomic>
#include <mutex>
#include <chrono>
#include <dlfcn.h>

#define HIP_TRY(expr)                          \
    do {                                       \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return GS2M_ERR_HIP; \
    } while (0)

namespace {

// getHigherMsb (rasterizer_impl.cu:31-44): number of key bits that cover the tile ids
uint32_t higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// Landing zones: mapped pinned words the GPU publishes a forward's counts into (common.h: GS2M_LAND_*), a ring of slots per
// device (never freed: the HIP runtime may already be gone when static destructors run).  A forward takes the next slot; the
// token it leaves behind (gs2m_raster_forward_token) names the slot and its generation, so that the dense-row count of THAT
// forward can be read later -- at backward time, from whichever thread autograd runs the backward on -- unless the slot has
// been reused since (64 forwards later: the caller then sizes for the worst case).
constexpr int kLandSlots = 64, kLandWords = 16;
struct LandRing {
    uint32_t* host = nullptr;  // kLandSlots x kLandWords words
    uint32_t* dev = nullptr;   // the same memory as the device sees it
    std::atomic<uint32_t> next{0};
    std::atomic<uint32_t> gen[kLandSlots];
};
LandRing g_land[64];
std::mutex g_land_mutex;
thread_local uint64_t t_last_token = 0;  // (device + 1) << 40 | generation << 8 | slot; 0: none
thread_local long long t_rows_hint = -1;  // gs2m_raster_backward_rows_hint: consumed by this thread's next backward
thread_local long long t_units_hint = -1, t_units_from_token = -1;  // the heavy-unit count that travels with it

// ---- optional per-stage timing with HIP events on the launch stream (bench.py) ----
// mode 0: off; 1: the two blend kernels only; 2: every stage; 3: the backward blend kernel only (an event pair costs
// ~6 us of stream bubble around the kernel it brackets: bench.py's timed region brackets the dominant kernel alone).
static_assert(GS2M_NUM_STAGES == 10, "stage table");
enum Stage { ST_PREPROCESS = 0, ST_UNUSED1, ST_SCAN, ST_EMIT, ST_TILE_SORT, ST_LISTS, ST_BLEND_FWD, ST_UNUSED7,
             ST_BLEND_BWD, ST_GAUSSIAN_BWD, ST_COUNT };
constexpr int kMaxRecords = 8192;
struct Prof {
    std::atomic<int> mode{0};  // read without the lock on the fast path (StageTimer), written under it
    int n = 0;
    hipEvent_t ev[kMaxRecords][2];
    int stage[kMaxRecords];
    int created = 0;
    int every = 1;   // mode 3: bracket every `every`-th launch of the backward blend only (gs2m_profile_sampling)
    int seen3 = 0;   // launches of the backward blend since the mode was set
};
Prof g_prof;
std::mutex g_prof_mutex;  // the record table is shared by every thread that calls into the library while profiling is on
// Mode switches: process-wide, read once at the top of a call (atomic: setting them from another thread is safe, and a
// call in flight keeps the values it started with).
std::atomic<int> g_reference_binning{0};
std::atomic<int> g_spin_wait{1};  // forward: poll the pinned num_rendered instead of hipStreamSynchronize
std::atomic<int> g_debug{0};      // gs2m_set_debug: synchronize + check after every stage
std::atomic<int> g_markers{0};    // gs2m_set_markers: roctx ranges around the stages

const char* const kStageNames[ST_COUNT] = {"preprocess", "-", "scan", "emit", "tile_sort", "depth_order+quad_lists", "blend_fwd", "-", "blend_bwd", "gaussian_bwd"};

// roctx ranges (rocprofv3 --marker-trace): resolved at run time so that the library has no link-time dependency
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
    void load() {
        if (tried) return;
        tried = true;
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
    }
};
Roctx g_roctx;

// One stage of a call: optional roctx range, optional HIP-event timing, and in debug mode a stream synchronize +
// error check when the stage ends, so that a fault is reported by the stage that caused it (`failed` = stage + 1).
struct StageTimer {
    hipStream_t s;
    int slot = -1;
    int stage;
    int* failed;
    bool marked = false;
    StageTimer(int stage_, hipStream_t s_, int* failed_ = nullptr) : s(s_), stage(stage_), failed(failed_) {
        if (g_markers.load(std::memory_order_relaxed)) {
            g_roctx.load();
            if (g_roctx.push) { g_roctx.push(kStageNames[stage]); marked = true; }
        }
        const bool blend = stage == ST_BLEND_FWD || stage == ST_BLEND_BWD;
        if (g_prof.mode.load(std::memory_order_relaxed) == 0) return;  // (the common case takes no lock)
        {
            std::lock_guard<std::mutex> lock(g_prof_mutex);
            if (g_prof.mode == 0 || (g_prof.mode == 1 && !blend) || (g_prof.mode == 3 && stage != ST_BLEND_BWD) || g_prof.n >= kMaxRecords) return;
            if (g_prof.mode == 3 && (g_prof.seen3++ % g_prof.every) != 0) return;
            slot = g_prof.n++;
            if (slot >= g_prof.created) {
                (void)hipEventCreate(&g_prof.ev[slot][0]);
                (void)hipEventCreate(&g_prof.ev[slot][1]);
                g_prof.created = slot + 1;
            }
            g_prof.stage[slot] = stage;
        }
        (void)hipEventRecord(g_prof.ev[slot][0], s);
    }
    ~StageTimer() {
        if (slot >= 0) (void)hipEventRecord(g_prof.ev[slot][1], s);
        if (failed && g_debug.load(std::memory_order_relaxed) && *failed == 0) {
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) *failed = stage + 1;
        }
        if (marked && g_roctx.pop) g_roctx.pop();
    }
};
#define DEBUG_CHECK()                                        \
    do {                                                     \
        if (failed_stage) return GS2M_ERR_STAGE(failed_stage - 1); \
    } while (0)

}  // namespace

extern "C" {

char* gs2m_prealloc_alloc(size_t bytes, void* user) {
    gs2m_prealloc* p = static_cast<gs2m_prealloc*>(user);
    if (!p) return nullptr;
    p->requested = bytes;
    if (p->ptr && bytes <= p->capacity) return p->ptr;
    p->used_fallback = 1;
    return p->fallback ? p->fallback(bytes, p->fallback_user) : nullptr;
}

const char* gs2m_version(void) { return "gs2m_raster 0.5 (gfx950, round 5)"; }

static int forward_impl(gs2m_alloc_fn geometry_alloc, void* geometry_user, gs2m_alloc_fn binning_alloc,
                        void* binning_user, gs2m_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D, const float* shs, const float* shs_rest,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* features,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                        float tan_fovy, int prefiltered, int feature_count, float* out_color, int* out_radii,
                        int* out_observe, float* out_buffer, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    const int reference_binning = g_reference_binning.load(), spin_wait = g_spin_wait.load();
    int failed_stage = 0;  // debug mode: 1 + the first stage whose kernels faulted
    if (P < 0 || width <= 0 || height <= 0 || feature_count < 0 || feature_count > GS2M_NUM_FEATURES) return GS2M_ERR_INVALID_ARG;
    if (!geometry_alloc || !binning_alloc || !image_alloc || !out_color || !out_buffer || !background) return GS2M_ERR_INVALID_ARG;
    if (P > 0 && (!means3D || !opacities || !out_radii || !out_observe || !viewmatrix || !projmatrix)) return GS2M_ERR_INVALID_ARG;
    if (P > 0 && ((shs == nullptr) == (colors_precomp == nullptr))) return GS2M_ERR_INVALID_ARG;
    if (P > 0 && (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr))) return GS2M_ERR_INVALID_ARG;
    if (P > 0 && shs && (D < 0 || D > 3 || M < (D + 1) * (D + 1) || !cam_pos)) return GS2M_ERR_INVALID_ARG;
    if (shs_rest && (!shs || M != 16 || (((uintptr_t)shs_rest) & 15))) return GS2M_ERR_UNSUPPORTED;  // split SH: M = 16 only
    if (P > 0 && feature_count > 0 && !features) return GS2M_ERR_INVALID_ARG;
    if (width > 16 * 65535 || height > 16 * 65535) return GS2M_ERR_UNSUPPORTED;
    if ((size_t)((width + GS2M_TILE - 1) / GS2M_TILE) * (size_t)((height + GS2M_TILE - 1) / GS2M_TILE) > ((size_t)1 << 28)) return GS2M_ERR_UNSUPPORTED;
    // the sorted values carry a 4-bit quadrant mask above the Gaussian id (binning.hip): ids stay below 2^28
    if (P >= (1 << GS2M_GID_BITS)) return GS2M_ERR_UNSUPPORTED;

    const int tiles_x = (width + GS2M_TILE - 1) / GS2M_TILE, tiles_y = (height + GS2M_TILE - 1) / GS2M_TILE;
    const size_t tiles = (size_t)tiles_x * tiles_y, N = (size_t)width * height;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);

    const size_t Pn = P > 0 ? (size_t)P : 1;
    GeomState gsz = gs2m_carve_geom(nullptr, Pn);
    char* gbase = geometry_alloc(gsz.total_bytes, geometry_user);
    if (!gbase) return GS2M_ERR_ALLOC;
    GeomState g = gs2m_carve_geom(gbase, Pn);

    ImageState isz = gs2m_carve_image(nullptr, N, tiles);
    char* ibase = image_alloc(isz.total_bytes, image_user);
    if (!ibase) return GS2M_ERR_ALLOC;
    ImageState im = gs2m_carve_image(ibase, N, tiles);

    // this call's landing slot
    int dev_id = 0;
    HIP_TRY(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 64) return GS2M_ERR_UNSUPPORTED;
    LandRing& ring = g_land[dev_id];
    if (!ring.host) {
        std::lock_guard<std::mutex> lock(g_land_mutex);
        if (!ring.host) {
            uint32_t* h = nullptr;
            HIP_TRY(hipHostMalloc((void**)&h, kLandSlots * kLandWords * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocPortable));
            HIP_TRY(hipHostGetDevicePointer((void**)&ring.dev, h, 0));
            for (int k = 0; k < kLandSlots; k++) ring.gen[k] = 0;
            ring.host = h;
        }
    }
    const uint32_t seq = ring.next.fetch_add(1), slot = seq % kLandSlots;
    const uint32_t generation = ring.gen[slot].fetch_add(1) + 1;
    volatile uint32_t* land = ring.host + slot * kLandWords;
    uint32_t* land_dev = ring.dev + slot * kLandWords;
    land[GS2M_LAND_R] = 0xFFFFFFFFu;  // a sentinel no count can take (R < 2^29)
    land[GS2M_LAND_HUNITS] = 0xFFFFFFFFu;  // (likewise: the two leave the GPU in one 8-byte store, the host still waits for both words)
    land[GS2M_LAND_PREFILTERED] = 0u;
    land[GS2M_LAND_ROWS] = 0u;
    t_last_token = ((uint64_t)(dev_id + 1) << 40) | ((uint64_t)(generation & 0xFFFFFFFFu) << 8) | slot;

    int R = 0;
    uint32_t U = 0, crowded = GS2M_CROWDED_WAVE;  // heavy units and the crowded-wave bar in force (common.h)
    if (P > 0) {
        {
            StageTimer t(ST_PREPROCESS, s, &failed_stage);
            // the preprocess kernel zeroes the digit histograms the emit kernel counts the tile sort's keys into
            ZeroJobs zj = {{g.tile_hist, nullptr, nullptr}, {(size_t)GS2M_HIST_COPIES * GS2M_HIST_COPY_WORDS, 0, 0}};
            gs2m_launch_preprocess(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, shs_rest, cov3D_precomp,
                                   colors_precomp, features, viewmatrix, projmatrix, cam_pos, width, height, tan_fovx,
                                   tan_fovy, focal_x, focal_y, tiles_x, tiles_y, out_radii, out_observe, g,
                                   reference_binning ? 0 : 1, zj, s);
        }
        // `prefiltered`: honoured as a checked promise (preprocess.hip); the check runs ahead of the kernel that publishes
        // num_rendered, so its flag has landed when the wait below returns
        if (prefiltered) gs2m_launch_prefiltered_check(P, means3D, viewmatrix, land_dev + GS2M_LAND_PREFILTERED, s);
        // The reference waits for num_rendered after its scan (rasterizer_impl.cu:269-270) and the GPU idles until the host
        // has seen the value, sized the binning buffer and launched the next kernel.  Here the value -- the sum of the
        // per-block counts the preprocess kernel left -- is the first thing the one-workgroup scan kernel publishes.
        {
            StageTimer t(ST_SCAN, s, &failed_stage);
            gs2m_launch_blockscan(P, g, land_dev, s);
        }
        HIP_TRY(hipGetLastError());
        DEBUG_CHECK();
        if (spin_wait) {
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            while (land[GS2M_LAND_R] == 0xFFFFFFFFu || land[GS2M_LAND_HUNITS] == 0xFFFFFFFFu) {
                __builtin_ia32_pause();
                if ((++spins & 0xFFFFu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            }
        }
        if (land[GS2M_LAND_R] == 0xFFFFFFFFu || land[GS2M_LAND_HUNITS] == 0xFFFFFFFFu) HIP_TRY(hipStreamSynchronize(s));  // polling disabled or timed out (e.g. a faulted stream)
        if (prefiltered && land[GS2M_LAND_PREFILTERED] != 0u) {  // a Gaussian behind the near plane: the reference traps the device here
            HIP_TRY(hipStreamSynchronize(s));  // nothing of this call is left in flight when the caller frees its buffers
            return GS2M_ERR_PREFILTERED;
        }
        // slots, list offsets, look-back words; gradient rows (4 per instance, heavy units padded to 64 instances) stay below the
        // GS2M_ROWS_BIG bit
        if (land[GS2M_LAND_R] >= (1u << 29)) return GS2M_ERR_UNSUPPORTED;
        R = (int)land[GS2M_LAND_R];
        U = land[GS2M_LAND_HUNITS];  // (published with num_rendered in one store)
        // The crowded-wave rule (common.h) repairs IMBALANCE between waves.  A frame whose Gaussians all cover a dozen tiles has every
        // wave crowded and nothing to repair -- but each of its Gaussians would reserve a unit of 256 rows for its dozen instances.
        // When the units ask for more than 6 rows per instance of the frame, they are counted again without the rule (Gaussians of
        // GS2M_HEAVY_TILES and more only: at most 64/40 x 4 = 6.4 rows per heavy instance) and the emit kernel is told.
        if ((size_t)U * 4 * GS2M_UNIT > (size_t)6 * (size_t)R + 65536) {
            crowded = GS2M_CROWDED_OFF;
            land[GS2M_LAND_R] = 0xFFFFFFFFu;
            land[GS2M_LAND_HUNITS] = 0xFFFFFFFFu;
            gs2m_launch_recount_heavy(P, g, s);
            gs2m_launch_blockscan(P, g, land_dev, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));  // (the rare path: no polling)
            if (land[GS2M_LAND_R] != (uint32_t)R) return GS2M_ERR_STAGE(ST_SCAN);
            U = land[GS2M_LAND_HUNITS];
        }
        if (U >= (1u << 22)) return GS2M_ERR_UNSUPPORTED;
    }

    const int tile_bits = (int)higher_msb((uint32_t)tiles);
    const size_t Rn = R > 0 ? (size_t)R : 1;
    const size_t btemp = gs2m_binning_temp_bytes(Rn, tile_bits);
    BinningState bsz = gs2m_carve_binning(nullptr, Rn, btemp, U);
    char* bbase = binning_alloc(bsz.total_bytes, binning_user);
    if (!bbase) return GS2M_ERR_ALLOC;
    BinningState b = gs2m_carve_binning(bbase, Rn, btemp, U);

    if (R > 0) {
        {   // the emit kernel also zeroes the tile sort's scratch and the tile ranges
            StageTimer t(ST_EMIT, s, &failed_stage);
            ZeroJobs zj = {{nullptr, im.ranges_raw, nullptr}, {0, tiles * 2, 0}};
            gs2m_radix_zero_region(b.temp, (size_t)R, tile_bits, &zj.p[0], &zj.words[0]);
            gs2m_launch_emit(P, width, height, tiles_x, tile_bits, g, b, U, crowded, land_dev, zj, s);
        }
        if (g_debug.load(std::memory_order_relaxed)) {  // debug mode: what the host was told against the emit kernel's own offsets
            uint32_t total = 0;
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(&total, g.counters, sizeof(total), hipMemcpyDeviceToHost));
            if (total != (uint32_t)R) return GS2M_ERR_STAGE(ST_EMIT);
        }
        {
            StageTimer t(ST_TILE_SORT, s, &failed_stage);
            // stable sort of (tile id, emission slot) on the tile bits: every tile's span comes out in index order; the last pass
            // also records every tile's range (identifyTileRanges, rasterizer_impl.cu:108-129)
            HIP_TRY(gs2m_radix_sort_pairs(b.temp, b.temp_bytes, b.keys_unsorted, nullptr, b.sort_keyA, b.sort_valA, b.tile_keys, b.slot_sorted,
                                          (size_t)R, tile_bits, true, s, SideSum{nullptr, nullptr, nullptr}, im.ranges_raw, g.tile_hist));
        }
    } else {
        HIP_TRY(gs2m_zero_async(im.ranges_raw, tiles * 2 * sizeof(uint32_t), s));
        land[GS2M_LAND_ROWS] = 1u;  // no instance, no row
        if (land[GS2M_LAND_HUNITS] == 0xFFFFFFFFu) land[GS2M_LAND_HUNITS] = 0u;  // (P == 0: nothing was launched)
    }
    DEBUG_CHECK();
    {
        StageTimer t(ST_LISTS, s, &failed_stage);  // per-tile (depth, id) order, ranges, the quadrant lists
        gs2m_launch_tile_sort(tiles, tiles_x, tiles_y, (size_t)(R > 0 ? R : 0), b, im, g, s);
    }
    {
        StageTimer t(ST_BLEND_FWD, s, &failed_stage);
        gs2m_launch_blend_fwd_q(width, height, tiles_x, tiles_y, feature_count, background, g, b, im, out_color, out_buffer, out_observe, s);
    }
    DEBUG_CHECK();
    HIP_TRY(hipGetLastError());
    return R;
}

static int backward_impl(int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* features, const float* viewmatrix, const float* projmatrix, const float* campos,
                         float tan_fovx, float tan_fovy, const int* radii, const float* buffer, char* geom_buffer,
                         char* binning_buffer, char* image_buffer, int feature_count, const float* grad_colors,
                         const float* grad_buffer, float* dL_dmeans2D, float* dL_dconics, float* dL_dopacities,
                         float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dshs, float* dL_dshs_rest, float* dL_dscales,
                         float* dL_drots, float* dL_dfeatures, gs2m_alloc_fn scratch_alloc, void* scratch_user,
                         void* stream_) {
    (void)buffer; (void)features;
    hipStream_t s = (hipStream_t)stream_;
    int failed_stage = 0;
    if (P == 0) return GS2M_OK;
    if (P >= (1 << GS2M_GID_BITS)) return GS2M_ERR_UNSUPPORTED;
    if (P < 0 || R < 0 || width <= 0 || height <= 0 || feature_count < 0 || feature_count > GS2M_NUM_FEATURES) return GS2M_ERR_INVALID_ARG;
    if (!geom_buffer || !binning_buffer || !image_buffer || !scratch_alloc || !grad_colors || !radii) return GS2M_ERR_INVALID_ARG;
    if (feature_count > 0 && !grad_buffer) return GS2M_ERR_INVALID_ARG;
    if (!dL_dmeans2D || !dL_dopacities || !dL_dmeans3D || !dL_dscales || !dL_drots || !dL_dfeatures) return GS2M_ERR_INVALID_ARG;
    // dL_dcolors / dL_dcov3D may be NULL when the corresponding input was not given (nobody reads them then)
    if ((colors_precomp && !dL_dcolors) || (cov3D_precomp && !dL_dcov3D)) return GS2M_ERR_INVALID_ARG;
    // dL_dshs (and dL_dshs_rest) may be NULL with SH input: dL/dSH is then not computed (a view whose colour gradient is identically zero)
    if (shs_rest && (M != 16 || ((dL_dshs == nullptr) != (dL_dshs_rest == nullptr)) || ((((uintptr_t)shs_rest) | ((uintptr_t)dL_dshs_rest)) & 15))) return GS2M_ERR_UNSUPPORTED;

    const int tiles_x = (width + GS2M_TILE - 1) / GS2M_TILE, tiles_y = (height + GS2M_TILE - 1) / GS2M_TILE;
    const size_t tiles = (size_t)tiles_x * tiles_y, N = (size_t)width * height;
    const size_t Rn = R > 0 ? (size_t)R : 1;
    GeomState g = gs2m_carve_geom(geom_buffer, (size_t)P);
    BinningState b = gs2m_carve_binning(binning_buffer, Rn, gs2m_binning_temp_bytes(Rn, (int)higher_msb((uint32_t)tiles)), 0);  // (no offset depends on the heavy units)
    ImageState im = gs2m_carve_image(image_buffer, N, tiles);

    // one partial-gradient row per (instance, quadrant), numbered densely over the view (binning.hip): `dense_rows` of them when
    // the caller passed the forward's count on (gs2m_raster_backward_rows_hint), 4 per instance at most otherwise
    const int rowf = gs2m_row_floats(feature_count);
    const long long hint = t_rows_hint;
    long long units = hint >= 0 ? t_units_hint : -1;
    t_rows_hint = -1;
    t_units_hint = -1;
    size_t nrows = 0;
    if (hint >= 0 && (size_t)hint <= Rn * 10 + 256) {
        nrows = (size_t)hint;
    } else if (R > 0) {  // no count from the caller: the forward left it on the device (a blocking read)
        uint32_t c[2] = {0u, 0u};
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(c, g.counters + GS2M_CNT_ROWS, sizeof(c), hipMemcpyDeviceToHost));
        nrows = c[0];
        units = c[1];
    }
    const size_t rows_bytes = gs2m_align_up((nrows > 0 ? nrows : 1) * (size_t)rowf * sizeof(float));
    char* sbase = scratch_alloc(rows_bytes + 2 * GS2M_ALIGN, scratch_user);
    if (!sbase) return GS2M_ERR_ALLOC;
    float* rows = (float*)gs2m_align_up((size_t)(uintptr_t)sbase);

    if (R > 0) {
        StageTimer t(ST_BLEND_BWD, s, &failed_stage);
        gs2m_launch_blend_bwd_q(width, height, tiles_x, tiles_y, feature_count, background, g, b, im, grad_colors,
                                grad_buffer, rows, s);
    }
    DEBUG_CHECK();
    {
        StageTimer tg(ST_GAUSSIAN_BWD, s, &failed_stage);  // row sums + the per-Gaussian chain: one kernel (+ the heavy units' sums ahead of it)
        if (R > 0) gs2m_launch_heavy_reduce(rows, rowf, b, g, units, s);
        gs2m_launch_gaussian_bwd(P, D, M, means3D, shs, shs_rest, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                                 viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy, radii, feature_count, g,
                                 rows, rowf, R > 0, dL_dmeans2D, dL_dconics, dL_dopacities, dL_dcolors, dL_dmeans3D,
                                 dL_dcov3D, dL_dshs, dL_dshs_rest, dL_dscales, dL_drots, dL_dfeatures, s);
    }
    DEBUG_CHECK();
    HIP_TRY(hipGetLastError());
    return GS2M_OK;
}

int gs2m_raster_forward(gs2m_alloc_fn geometry_alloc, void* geometry_user, gs2m_alloc_fn binning_alloc,
                        void* binning_user, gs2m_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* features,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                        float tan_fovy, int prefiltered, int feature_count, float* out_color, int* out_radii,
                        int* out_observe, float* out_buffer, void* stream_) {
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background, width, height, means3D, shs, nullptr, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, features, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, feature_count, out_color, out_radii, out_observe, out_buffer, stream_);
}

int gs2m_raster_forward_split_sh(gs2m_alloc_fn geometry_alloc, void* geometry_user, gs2m_alloc_fn binning_alloc,
                        void* binning_user, gs2m_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D, const float* sh_dc, const float* sh_rest,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* features,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                        float tan_fovy, int prefiltered, int feature_count, float* out_color, int* out_radii,
                        int* out_observe, float* out_buffer, void* stream_) {
    if (!sh_dc || !sh_rest) return GS2M_ERR_INVALID_ARG;
    return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background, width, height, means3D, sh_dc, sh_rest, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, features, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, feature_count, out_color, out_radii, out_observe, out_buffer, stream_);
}

int gs2m_raster_backward(int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* features, const float* viewmatrix, const float* projmatrix, const float* campos,
                         float tan_fovx, float tan_fovy, const int* radii, const float* buffer, char* geom_buffer,
                         char* binning_buffer, char* image_buffer, int feature_count, const float* grad_colors,
                         const float* grad_buffer, float* dL_dmeans2D, float* dL_dconics, float* dL_dopacities,
                         float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dshs, float* dL_dscales,
                         float* dL_drots, float* dL_dfeatures, gs2m_alloc_fn scratch_alloc, void* scratch_user,
                         void* stream_) {
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, nullptr, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, features, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, buffer, geom_buffer, binning_buffer, image_buffer, feature_count, grad_colors, grad_buffer, dL_dmeans2D, dL_dconics, dL_dopacities, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dshs, nullptr, dL_dscales, dL_drots, dL_dfeatures, scratch_alloc, scratch_user, stream_);
}

int gs2m_raster_backward_split_sh(int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* sh_dc, const float* sh_rest, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* features, const float* viewmatrix, const float* projmatrix, const float* campos,
                         float tan_fovx, float tan_fovy, const int* radii, const float* buffer, char* geom_buffer,
                         char* binning_buffer, char* image_buffer, int feature_count, const float* grad_colors,
                         const float* grad_buffer, float* dL_dmeans2D, float* dL_dconics, float* dL_dopacities,
                         float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscales,
                         float* dL_drots, float* dL_dfeatures, gs2m_alloc_fn scratch_alloc, void* scratch_user,
                         void* stream_) {
    if (!sh_dc || !sh_rest) return GS2M_ERR_INVALID_ARG;
    return backward_impl(P, D, M, R, background, width, height, means3D, sh_dc, sh_rest, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, features, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, buffer, geom_buffer, binning_buffer, image_buffer, feature_count, grad_colors, grad_buffer, dL_dmeans2D, dL_dconics, dL_dopacities, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh_dc, dL_dsh_rest, dL_dscales, dL_drots, dL_dfeatures, scratch_alloc, scratch_user, stream_);
}

int gs2m_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                             uint8_t* present, void* stream_) {
    (void)projmatrix;
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if (P == 0) return GS2M_OK;
    if (!means3D || !viewmatrix || !present) return GS2M_ERR_INVALID_ARG;
    gs2m_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream_);
    HIP_TRY(hipGetLastError());
    return GS2M_OK;
}

int gs2m_set_reference_binning(int on) {
    g_reference_binning = on ? 1 : 0;
    return GS2M_OK;
}

int gs2m_set_tile_sort_policy(int policy) {
    if (policy < 0 || policy > 3) return GS2M_ERR_INVALID_ARG;
    gs2m_set_tile_sort_policy_impl(policy);
    return GS2M_OK;
}

int gs2m_set_debug(int on) {
    g_debug = on ? 1 : 0;
    return GS2M_OK;
}

int gs2m_set_markers(int on) {
    g_markers = on ? 1 : 0;
    if (on) {
        g_roctx.load();
        if (!g_roctx.push || !g_roctx.pop) return GS2M_ERR_UNSUPPORTED;  // libroctx64 not found
    }
    return GS2M_OK;
}

const char* gs2m_stage_name(int stage) { return stage >= 0 && stage < ST_COUNT ? kStageNames[stage] : "?"; }

int gs2m_set_spin_wait(int on) {
    g_spin_wait = on ? 1 : 0;
    return GS2M_OK;
}

int gs2m_profile_mode(int mode) {
    if (mode < 0 || mode > 3) return GS2M_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof.mode = mode;
    g_prof.n = 0;
    g_prof.seen3 = 0;
    return GS2M_OK;
}

int gs2m_profile_sampling(int every) {
    if (every < 1) return GS2M_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof.every = every;
    return GS2M_OK;
}

int gs2m_profile_collect(float* stage_ms, int* stage_count, int n_stages) {
    if (!stage_ms || !stage_count || n_stages < GS2M_NUM_STAGES) return GS2M_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (int i = 0; i < n_stages; i++) { stage_ms[i] = 0.f; stage_count[i] = 0; }
    for (int r = 0; r < g_prof.n; r++) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(g_prof.ev[r][1]));
        HIP_TRY(hipEventElapsedTime(&ms, g_prof.ev[r][0], g_prof.ev[r][1]));
        stage_ms[g_prof.stage[r]] += ms;
        stage_count[g_prof.stage[r]] += 1;
    }
    g_prof.n = 0;
    return GS2M_OK;
}

int gs2m_debug_layout(int P, int R, int width, int height, gs2m_layout* out) {
    if (!out || P < 0 || R < 0 || width <= 0 || height <= 0) return GS2M_ERR_INVALID_ARG;
    const int tiles_x = (width + GS2M_TILE - 1) / GS2M_TILE, tiles_y = (height + GS2M_TILE - 1) / GS2M_TILE;
    const size_t tiles = (size_t)tiles_x * tiles_y, N = (size_t)width * height;
    const size_t Pn = P > 0 ? (size_t)P : 1, Rn = R > 0 ? (size_t)R : 1;
    GeomState g = gs2m_carve_geom(nullptr, Pn);
    BinningState b = gs2m_carve_binning(nullptr, Rn, gs2m_binning_temp_bytes(Rn, (int)higher_msb((uint32_t)tiles)), 0);
    ImageState im = gs2m_carve_image(nullptr, N, tiles);
    out->geom_bytes = g.total_bytes;
    out->rec = (uint64_t)(uintptr_t)g.rec;
    out->tiles_touched = (uint64_t)(uintptr_t)g.tiles_touched;
    out->depth_key = (uint64_t)(uintptr_t)g.depth_key;
    out->rect = (uint64_t)(uintptr_t)g.rect;
    out->gauss_rows = (uint64_t)(uintptr_t)g.gauss_rows;
    out->clamped = (uint64_t)(uintptr_t)g.clamped;
    out->wave_rowbase = (uint64_t)(uintptr_t)g.wave_rowbase;
    out->counters = (uint64_t)(uintptr_t)g.counters;
    out->binning_bytes = b.total_bytes;
    out->point_list = (uint64_t)(uintptr_t)b.point_list;
    out->tile_keys = (uint64_t)(uintptr_t)b.tile_keys;
    out->qlist = (uint64_t)(uintptr_t)b.qlist;
    out->qrow = (uint64_t)(uintptr_t)b.qrow;
    out->image_bytes = im.total_bytes;
    out->final_T = (uint64_t)(uintptr_t)im.final_T;
    out->n_contrib = (uint64_t)(uintptr_t)im.n_contrib;
    out->ranges = (uint64_t)(uintptr_t)im.ranges;
    out->qcount = (uint64_t)(uintptr_t)im.qcount;
    return GS2M_OK;
}

unsigned long long gs2m_raster_forward_token(void) { return (unsigned long long)t_last_token; }

long long gs2m_raster_dense_rows(unsigned long long token) {
    if (token == 0ull) return -1;
    const int dev = (int)(token >> 40) - 1;
    const uint32_t slot = (uint32_t)(token & 0xFFull), generation = (uint32_t)((token >> 8) & 0xFFFFFFFFull);
    if (dev < 0 || dev >= 64 || slot >= (uint32_t)kLandSlots || !g_land[dev].host) return -1;
    LandRing& ring = g_land[dev];
    volatile uint32_t* land = ring.host + slot * kLandWords;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    for (;;) {
        if (ring.gen[slot].load() != generation) return -1;  // the slot has been handed to a later forward
        const uint32_t v = land[GS2M_LAND_ROWS];
        if (v != 0u) {
            t_units_from_token = (long long)land[GS2M_LAND_HUNITS];
            return ring.gen[slot].load() == generation ? (long long)v - 1 : -1;
        }
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return -1;
    }
}

int gs2m_raster_backward_rows_hint(long long dense_rows) {
    t_rows_hint = dense_rows;
    t_units_hint = dense_rows >= 0 ? t_units_from_token : -1;  // (the forward's heavy-unit count came back with its row count)
    t_units_from_token = -1;
    return GS2M_OK;
}

// Test hook: tile_sort.hip on caller-made spans (no rasterization): per tile {~first, last + 1} as the tile sort records them, the
// emission slots of every span in index order, {id | mask, relative row, depth key, -} per slot and a per-wave row base
// table -> ranges, sorted values, the four quadrant lists and their rows and counts.
int gs2m_debug_tile_sort(int tiles, const unsigned* ranges_raw, unsigned* ranges, const unsigned* slot_sorted, const unsigned* e_rec,
                         const unsigned* wave_rowbase, unsigned* point_list, unsigned* row_tmp, unsigned* qlist,
                         unsigned* qrow, unsigned* qcount, void* stream_) {
    if (tiles < 0 || !ranges_raw || !ranges || !slot_sorted || !e_rec || !wave_rowbase || !point_list || !row_tmp || !qlist || !qrow || !qcount) return GS2M_ERR_INVALID_ARG;
    BinningState b = {};
    b.slot_sorted = const_cast<uint32_t*>(slot_sorted); b.e_rec = reinterpret_cast<uint4*>(const_cast<unsigned*>(e_rec));
    b.point_list = point_list; b.sort_valA = row_tmp; b.qlist = reinterpret_cast<uint2*>(qlist); b.qrow = qrow;
    ImageState im = {};
    im.ranges_raw = const_cast<uint32_t*>(ranges_raw);
    im.ranges = reinterpret_cast<uint2*>(ranges);
    im.qcount = qcount;
    GeomState g = {};
    g.wave_rowbase = const_cast<uint32_t*>(wave_rowbase);
    // the span-class words the sort's kernels pass on to each other (common.h: GS2M_CNT_SPAN_*): zeroed per call, as blockscan_kernel does
    static uint32_t* dbg_counters[64] = {nullptr};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return GS2M_ERR_UNSUPPORTED;
    if (dbg_counters[dev] == nullptr) HIP_TRY(hipMalloc(&dbg_counters[dev], 64 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(dbg_counters[dev], 0, 64 * sizeof(uint32_t), (hipStream_t)stream_));
    g.counters = dbg_counters[dev];
    gs2m_launch_tile_sort((size_t)tiles, tiles, 1, SIZE_MAX, b, im, g, (hipStream_t)stream_);  // (a one-row tile grid; the number of entries is not known here)
    HIP_TRY(hipGetLastError());
    return GS2M_OK;
}

}  // extern "C"
