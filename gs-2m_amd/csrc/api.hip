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

// pinned landing zone for num_rendered, one per host thread (never freed: the HIP runtime
// may already be gone when thread-local destructors run)
struct Pinned {
    uint32_t* p = nullptr;    // host pointer of the pinned landing zone
    uint32_t* dev = nullptr;  // the same memory as the device sees it
    uint32_t* acc[256] = {};  // per device: two zeroed words the side sum of the histogram kernel works in
};
thread_local Pinned t_pinned;

// ---- optional per-stage timing with HIP events on the launch stream (bench.py) ----
// mode 0: off; 1: the two blend kernels only; 2: every stage; 3: the backward blend kernel only (an event pair costs
// ~6 us of stream bubble around the kernel it brackets: bench.py's timed region brackets the dominant kernel alone).
static_assert(GS2M_NUM_STAGES == 10, "stage table");
enum Stage { ST_PREPROCESS = 0, ST_DEPTH_SORT, ST_SCAN, ST_EMIT, ST_TILE_SORT, ST_RANGES, ST_BLEND_FWD, ST_OBSERVE,
             ST_BLEND_BWD, ST_GAUSSIAN_BWD, ST_COUNT };
constexpr int kMaxRecords = 8192;
struct Prof {
    int mode = 0;
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

const char* const kStageNames[ST_COUNT] = {"preprocess", "depth_sort", "scan", "emit", "tile_sort", "ranges+quad_lists",
                                           "blend_fwd", "observe", "blend_bwd", "gaussian_bwd"};

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
        if (g_prof.mode == 0) return;  // (the common case takes no lock)
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
    if (p->ptr && bytes <= p->capacity) return p->ptr;
    p->used_fallback = 1;
    return p->fallback ? p->fallback(bytes, p->fallback_user) : nullptr;
}

const char* gs2m_version(void) { return "gs2m_raster 0.3 (gfx950, round 3)"; }

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
    // the sorted values carry a 4-bit quadrant mask above the Gaussian id (binning.hip): ids stay below 2^28
    if (P >= (1 << GS2M_GID_BITS)) return GS2M_ERR_UNSUPPORTED;

    const int tiles_x = (width + GS2M_TILE - 1) / GS2M_TILE, tiles_y = (height + GS2M_TILE - 1) / GS2M_TILE;
    const size_t tiles = (size_t)tiles_x * tiles_y, N = (size_t)width * height;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);

    const size_t Pn = P > 0 ? (size_t)P : 1;
    const size_t gtemp = gs2m_geom_temp_bytes(Pn);
    GeomState gsz = gs2m_carve_geom(nullptr, Pn, gtemp);
    char* gbase = geometry_alloc(gsz.total_bytes, geometry_user);
    if (!gbase) return GS2M_ERR_ALLOC;
    GeomState g = gs2m_carve_geom(gbase, Pn, gtemp);

    ImageState isz = gs2m_carve_image(nullptr, N, tiles);
    char* ibase = image_alloc(isz.total_bytes, image_user);
    if (!ibase) return GS2M_ERR_ALLOC;
    ImageState im = gs2m_carve_image(ibase, N, tiles);

    int R = 0;
    uint32_t* tile_hist = nullptr;  // the tile sort's digit histograms (zeroed by the preprocess kernel, filled by the emit kernel)
    uint32_t *block_sums = nullptr, *super_sums = nullptr;  // tiles_touched summed over blocks of 256 (and of 65536) depth-sorted Gaussians (filled by the depth sort's last pass)
    if (P > 0) {
        // scratch of the depth sort and of the scan, side by side in g.temp; zeroed by the preprocess kernel
        char* sort_temp = g.temp;
        const size_t sort_temp_bytes = gs2m_align_up(gs2m_radix_temp_bytes((size_t)P, 32));
        char* front_temp = g.temp + sort_temp_bytes;  // block sums of tiles_touched + the tile sort's digit histograms
        if (!t_pinned.p) {
            HIP_TRY(hipHostMalloc((void**)&t_pinned.p, 64, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void**)&t_pinned.dev, t_pinned.p, 0));
        }
        int dev_id = 0;
        HIP_TRY(hipGetDevice(&dev_id));
        uint32_t* acc = nullptr;  // per device: the words the side sum of the histogram kernel works in
        if (dev_id >= 0 && dev_id < 256) {
            if (!t_pinned.acc[dev_id]) HIP_TRY(hipMalloc((void**)&t_pinned.acc[dev_id], 64));
            acc = t_pinned.acc[dev_id];
        }
        // zeroed by the preprocess kernel, on this stream, ahead of every use: the depth sort's scratch, the scan's, and the
        // side sum's accumulator (so a call that died half way cannot leave a count behind for the next one)
        ZeroJobs zj = {{nullptr, nullptr, acc}, {0, 0, acc ? (size_t)4 : (size_t)0}};
        gs2m_radix_zero_region(sort_temp, (size_t)P, 32, &zj.p[0], &zj.words[0]);
        gs2m_front_zero_region(front_temp, (size_t)P, &zj.p[1], &zj.words[1]);
        tile_hist = gs2m_tile_hist_ptr(front_temp, (size_t)P);
        block_sums = gs2m_block_sums_ptr(front_temp);
        super_sums = gs2m_super_sums_ptr(front_temp, (size_t)P);
        {
            StageTimer t(ST_PREPROCESS, s, &failed_stage);
            gs2m_launch_preprocess(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, shs_rest, cov3D_precomp,
                                   colors_precomp, features, viewmatrix, projmatrix, cam_pos, width, height, tan_fovx,
                                   tan_fovy, focal_x, focal_y, tiles_x, tiles_y, out_radii, out_observe, g,
                                   reference_binning ? 0 : 1, zj, s);
        }
        // The reference waits for num_rendered after its scan (rasterizer_impl.cu:269-270) and the GPU idles until the
        // host has seen the value, sized the binning buffer and launched the next kernel.  Here the value -- the plain sum
        // of tiles_touched, whatever the order -- is added up on the side by the depth sort's histogram kernel, the first
        // kernel behind the preprocessing, and its last workgroup stores it into a mapped pinned word (one aligned
        // system-scope 32-bit store; a 4-byte hipMemcpyAsync may be carried out byte by byte: torn counts were seen).
        // The host polls that word (a sentinel no count can take: R < 2^30) while the sort passes still
        // run, and has the binning kernels queued behind them before they finish: no idle gap.
        volatile uint32_t* land = t_pinned.p;
        land[0] = 0xFFFFFFFFu;
        land[1] = 0u;
        // `prefiltered`: honoured as a checked promise (preprocess.hip); the check runs ahead of the histogram kernel whose
        // last workgroup publishes num_rendered, so its flag has landed when the wait below returns
        if (prefiltered) gs2m_launch_prefiltered_check(P, means3D, viewmatrix, t_pinned.dev + 1, s);
        {   // 1. depth order of the Gaussians themselves (stable: ties keep id order)
            StageTimer t(ST_DEPTH_SORT, s, &failed_stage);
            const SideSum sum = {acc ? g.tiles_touched : nullptr, acc, t_pinned.dev};
            // the last pass also leaves the sums of tiles_touched over blocks of 256 sorted Gaussians (the emit kernel's prefix)
            HIP_TRY(gs2m_radix_sort_pairs(sort_temp, sort_temp_bytes, g.depth_key, nullptr, g.sort_keyA, g.sort_valA,
                                          g.depth_key_sorted, g.sorted_gid, (size_t)P, 32, true, s, sum, nullptr, nullptr,
                                          SideBuckets{g.tiles_touched, block_sums, super_sums}));
        }
        HIP_TRY(hipGetLastError());
        DEBUG_CHECK();
        if (spin_wait) {
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            while (land[0] == 0xFFFFFFFFu) {
                __builtin_ia32_pause();
                if ((++spins & 0xFFFFu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            }
        }
        if (land[0] == 0xFFFFFFFFu) HIP_TRY(hipStreamSynchronize(s));  // polling disabled or timed out (e.g. a faulted stream)
        if (prefiltered && land[1] != 0u) {  // a Gaussian behind the near plane: the reference traps the device here
            HIP_TRY(hipStreamSynchronize(s));  // nothing of this call is left in flight when the caller frees its buffers
            return GS2M_ERR_PREFILTERED;
        }
        if (land[0] >= (1u << 30)) return GS2M_ERR_UNSUPPORTED;  // the look-back status words carry 30 value bits
        R = (int)land[0];
    }

    const int tile_bits = (int)higher_msb((uint32_t)tiles);
    const size_t Rn = R > 0 ? (size_t)R : 1;
    const size_t btemp = gs2m_binning_temp_bytes(Rn, tile_bits);
    BinningState bsz = gs2m_carve_binning(nullptr, Rn, btemp);
    char* bbase = binning_alloc(bsz.total_bytes, binning_user);
    if (!bbase) return GS2M_ERR_ALLOC;
    BinningState b = gs2m_carve_binning(bbase, Rn, btemp);

    if (R > 0) {
        {   // the emit kernel also zeroes the tile sort's scratch, the tile ranges and the per-instance observe counts
            StageTimer t(ST_EMIT, s, &failed_stage);
            ZeroJobs zj = {{nullptr, nullptr, im.ranges_raw}, {0, 0, tiles * 2}};
            gs2m_radix_zero_region(b.temp, (size_t)R, tile_bits, &zj.p[0], &zj.words[0]);
            gs2m_launch_emit(P, width, height, tiles_x, tile_bits, tile_hist, block_sums, super_sums, g, b, zj, s);
        }
        if (g_debug.load(std::memory_order_relaxed)) {  // debug mode: the histogram kernel's side sum against the emit kernel's own prefix-sum total
            uint32_t total = 0;
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(&total, g.counters, sizeof(total), hipMemcpyDeviceToHost));
            if (total != (uint32_t)R) return GS2M_ERR_STAGE(ST_EMIT);
        }
        {
            StageTimer t(ST_TILE_SORT, s, &failed_stage);
            // the last pass also records every tile's range (identifyTileRanges, rasterizer_impl.cu:108-129)
            HIP_TRY(gs2m_radix_sort_pairs(b.temp, b.temp_bytes, b.keys_unsorted, b.vals_unsorted, b.sort_keyA, b.sort_valA,
                                          b.tile_keys, b.point_list, (size_t)R, tile_bits, true, s, SideSum{nullptr, nullptr, nullptr},
                                          im.ranges_raw, tile_hist));
        }
    } else {
        HIP_TRY(gs2m_zero_async(im.ranges_raw, tiles * 2 * sizeof(uint32_t), s));
    }
    DEBUG_CHECK();
    {
        StageTimer t(ST_RANGES, s, &failed_stage);  // second binning level: counted with the ranges stage
        gs2m_launch_quad_lists(width, height, tiles_x, tiles_y, g, b, im, s);
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
    if (shs && M > 0 && !dL_dshs) return GS2M_ERR_INVALID_ARG;
    if (shs_rest && (M != 16 || !dL_dshs_rest || ((((uintptr_t)shs_rest) | ((uintptr_t)dL_dshs_rest)) & 15))) return GS2M_ERR_UNSUPPORTED;

    const int tiles_x = (width + GS2M_TILE - 1) / GS2M_TILE, tiles_y = (height + GS2M_TILE - 1) / GS2M_TILE;
    const size_t tiles = (size_t)tiles_x * tiles_y, N = (size_t)width * height;
    const size_t Rn = R > 0 ? (size_t)R : 1;
    GeomState g = gs2m_carve_geom(geom_buffer, (size_t)P, gs2m_geom_temp_bytes((size_t)P));
    BinningState b = gs2m_carve_binning(binning_buffer, Rn, gs2m_binning_temp_bytes(Rn, (int)higher_msb((uint32_t)tiles)));
    ImageState im = gs2m_carve_image(image_buffer, N, tiles);

    // one partial-gradient row per (instance, quadrant) at most, numbered densely (binning.hip), and one reduced row
    // per Gaussian
    const int rowf = gs2m_row_floats(feature_count);
    const size_t rows_bytes = gs2m_align_up(Rn * 4 * (size_t)rowf * sizeof(float));
    const size_t sums_bytes = gs2m_align_up((size_t)(P > 0 ? P : 1) * rowf * sizeof(float));
    char* sbase = scratch_alloc(rows_bytes + sums_bytes + 2 * GS2M_ALIGN, scratch_user);
    if (!sbase) return GS2M_ERR_ALLOC;
    char* al = (char*)gs2m_align_up((size_t)(uintptr_t)sbase);
    float* rows = (float*)al;
    float* sums = (float*)(al + rows_bytes);

    if (R > 0) {
        StageTimer t(ST_BLEND_BWD, s, &failed_stage);
        gs2m_launch_blend_bwd_q(width, height, tiles_x, tiles_y, feature_count, background, g, b, im, grad_colors,
                                grad_buffer, rows, s);
    }
    DEBUG_CHECK();
    {
        StageTimer tg(ST_GAUSSIAN_BWD, s, &failed_stage);
        if (R > 0) gs2m_launch_row_reduce_dense(P, g, rows, rowf, sums, s);
        else HIP_TRY(gs2m_zero_async(sums, (size_t)P * rowf * sizeof(float), s));
        gs2m_launch_gaussian_bwd(P, D, M, means3D, shs, shs_rest, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                                 viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy, radii, feature_count, g,
                                 sums, rowf, dL_dmeans2D, dL_dconics, dL_dopacities, dL_dcolors, dL_dmeans3D,
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
    GeomState g = gs2m_carve_geom(nullptr, Pn, gs2m_geom_temp_bytes(Pn));
    BinningState b = gs2m_carve_binning(nullptr, Rn, gs2m_binning_temp_bytes(Rn, (int)higher_msb((uint32_t)tiles)));
    ImageState im = gs2m_carve_image(nullptr, N, tiles);
    out->geom_bytes = g.total_bytes;
    out->rec = (uint64_t)(uintptr_t)g.rec;
    out->tiles_touched = (uint64_t)(uintptr_t)g.tiles_touched;
    out->depth_key = (uint64_t)(uintptr_t)g.depth_key;
    out->sorted_gid = (uint64_t)(uintptr_t)g.sorted_gid;
    out->sorted_off = (uint64_t)(uintptr_t)g.sorted_off;
    out->clamped = (uint64_t)(uintptr_t)g.clamped;
    out->binning_bytes = b.total_bytes;
    out->point_list = (uint64_t)(uintptr_t)b.point_list;
    out->tile_keys = (uint64_t)(uintptr_t)b.tile_keys;
    out->inst_obs = (uint64_t)(uintptr_t)b.inst_obs;
    out->image_bytes = im.total_bytes;
    out->final_T = (uint64_t)(uintptr_t)im.final_T;
    out->n_contrib = (uint64_t)(uintptr_t)im.n_contrib;
    out->ranges = (uint64_t)(uintptr_t)im.ranges;
    return GS2M_OK;
}

}  // extern "C"
