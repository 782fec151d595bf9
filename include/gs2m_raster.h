/*
 * gs2m_raster.h -- C ABI of the MI355X-native (gfx950) GS-2M rasterizer hot path.
 *
 * Drop-in boundary: these entry points take exactly what the reference's native seam
 * takes (raw device pointers, sizes, allocator callbacks) so that a binding written
 * against CudaRasterizer::Rasterizer can bind them one for one.  Reference interface
 * replaced (paths relative to /root/reference/submodules/):
 *
 *   gs2m_raster_forward       <- CudaRasterizer::Rasterizer::forward
 *                                diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:31-59
 *                                (impl rasterizer_impl.cu:185-330; torch caller rasterize_points.cu:30-113)
 *   gs2m_raster_backward      <- CudaRasterizer::Rasterizer::backward
 *                                cuda_rasterizer/rasterizer.h:61-88 (impl rasterizer_impl.cu:334-438;
 *                                torch caller rasterize_points.cu:115-200)
 *   gs2m_raster_mark_visible  <- CudaRasterizer::Rasterizer::markVisible
 *                                cuda_rasterizer/rasterizer.h:23-29 (impl rasterizer_impl.cu:132-143)
 *   gs2m_knn_dist2            <- SimpleKNN::knn  simple-knn/simple_knn.h, simple_knn.cu:169-204
 *                                (torch caller spatial.cu:15-24, python name distCUDA2)
 *
 * Differences from the reference seam, all additive:
 *   - every call takes the HIP stream to launch on (the reference uses the default stream);
 *   - allocator callbacks are plain function pointers + a user pointer instead of
 *     std::function<char*(size_t)>;
 *   - backward takes one extra scratch allocator (per tile-instance partial-gradient rows:
 *     the reference accumulates with float atomics and needs no scratch);
 *   - int status / num_rendered return codes instead of C++ exceptions.
 *
 * All pointers are DEVICE pointers to contiguous fp32 / int32 data unless stated.  Null
 * pointers select the same alternatives as in the reference (shs vs colors_precomp,
 * scales+rotations vs cov3D_precomp).  Output tensors out_color (3,H,W), out_radii (P),
 * out_observe (P), out_buffer (10,H,W) and all gradient tensors must be allocated by the
 * caller; they do NOT have to be zero-filled (the kernels write every element).
 * The three scratch buffers returned by the forward callbacks must be kept alive and
 * passed unchanged to backward; their layout is private to this library.
 */
#ifndef GS2M_RASTER_H
#define GS2M_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS2M_NUM_CHANNELS 3  /* cuda_rasterizer/config.h:15 */
#define GS2M_NUM_FEATURES 10 /* cuda_rasterizer/config.h:16 */
#define GS2M_TILE 16         /* cuda_rasterizer/config.h:17-18 */

/* error codes (negative returns) */
#define GS2M_OK 0
#define GS2M_ERR_INVALID_ARG (-1)
#define GS2M_ERR_HIP (-2)
#define GS2M_ERR_ALLOC (-3)
#define GS2M_ERR_UNSUPPORTED (-4)
/* `prefiltered` was set and a Gaussian lies behind the near plane (view z <= 0.2): the reference prints "Point culled!
 * This point should have been prefiltered" and TRAPS the device there (cuda_rasterizer/auxiliary.h:155-158); this
 * library reports it through the forward's return value instead and leaves the context usable. */
#define GS2M_ERR_PREFILTERED (-5)
/* debug mode (gs2m_set_debug): the kernels of pipeline stage `st` (order below, GS2M_NUM_STAGES) faulted */
#define GS2M_ERR_STAGE(st) (-100 - (st))

/* Scratch allocator: must return a DEVICE pointer to at least `bytes` bytes (any
 * alignment; the library aligns internally) that stays valid until the matching
 * backward has run.  Mirrors resizeFunctional, rasterize_points.cu:22-28. */
typedef char* (*gs2m_alloc_fn)(size_t bytes, void* user);

/* Returns num_rendered (>= 0, number of Gaussian/tile instances) or a negative error. */
int gs2m_raster_forward(
    gs2m_alloc_fn geometry_alloc, void* geometry_user,
    gs2m_alloc_fn binning_alloc, void* binning_user,
    gs2m_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* features,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    int feature_count,
    float* out_color,
    int* out_radii,
    int* out_observe,
    float* out_buffer,
    void* stream);

/* Returns GS2M_OK or a negative error.  dL_dconics may be NULL (it is an internal
 * temporary in the reference, rasterize_points.cu:154); dL_dcolors may be NULL when colors_precomp is NULL and dL_dcov3D
 * when cov3D_precomp is NULL (the reference writes them regardless and its Python side drops them: 36 bytes per Gaussian
 * of HBM writes nobody reads). */
int gs2m_raster_backward(
    int P, int D, int M, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* features,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    const float* buffer, /* unused, as in the reference (backward.cu:428) */
    char* geom_buffer,
    char* binning_buffer,
    char* image_buffer,
    int feature_count,
    const float* grad_colors,
    const float* grad_buffer,
    float* dL_dmeans2D,   /* (P,4) */
    float* dL_dconics,    /* (P,2,2) or NULL */
    float* dL_dopacities, /* (P,1) */
    float* dL_dcolors,    /* (P,3), or NULL without colors_precomp */
    float* dL_dmeans3D,   /* (P,3) */
    float* dL_dcov3D,     /* (P,6), or NULL without cov3D_precomp */
    float* dL_dshs,       /* (P,M,3); NULL: dL/dSH is not wanted (a view whose colour gradient is identically zero: the 48 stores per Gaussian are skipped) */
    float* dL_dscales,    /* (P,3) */
    float* dL_drots,      /* (P,4) */
    float* dL_dfeatures,  /* (P,10) */
    gs2m_alloc_fn scratch_alloc, void* scratch_user,
    void* stream);

/* present: P bytes (bool). */
int gs2m_raster_mark_visible(int P, const float* means3D, const float* viewmatrix,
                             const float* projmatrix, uint8_t* present, void* stream);

/* points (P,3) fp32 -> mean_dists (P) fp32: mean squared distance to the 3 nearest
 * other points.  scratch_alloc provides temporary device memory. */
int gs2m_knn_dist2(int P, const float* points, float* mean_dists,
                   gs2m_alloc_fn scratch_alloc, void* scratch_user, void* stream);

/* ---- introspection used by the parity tests (not part of the reference seam) ---- */

/* Byte offsets of the private arrays inside the three scratch buffers, relative to the
 * 256-byte-aligned base of each buffer.  Filled by gs2m_debug_layout. */
typedef struct gs2m_layout {
    /* geometry buffer */
    uint64_t geom_bytes;
    uint64_t rec;           /* P x 32 floats: blend records, see DESIGN.md */
    uint64_t tiles_touched; /* P u32 */
    uint64_t depth_key;     /* P u32: fp32 bits of view-space depth, 0xFFFFFFFF if culled */
    uint64_t rect;          /* P x uint2: emitted tile rectangle {min x | min y << 16, w | h << 16} */
    uint64_t gauss_rows;    /* P u32: partial-gradient rows of the Gaussian (bit 31: summed by a whole workgroup) */
    uint64_t clamped;       /* P u8 bit mask (bit c = SH channel c clamped) */
    uint64_t wave_rowbase;  /* ceil(P / 64) u32: first gradient row of every wave of 64 consecutive Gaussians */
    uint64_t counters;      /* 64 u32: [0] num_rendered, [1] the same from the block sums, [2] dense gradient rows */
    /* binning buffer */
    uint64_t binning_bytes;
    uint64_t point_list;    /* R u32: Gaussian ids (| quadrant mask << 28) sorted by (tile, depth, id) */
    uint64_t tile_keys;     /* R u32: tile id of each sorted instance */
    uint64_t qlist;         /* 4R x uint2: the per-(tile, quadrant) lists {id | mask << 28, position in the tile list} */
    uint64_t qrow;          /* 4R u32: gradient row of each list entry */
    /* image buffer */
    uint64_t image_bytes;
    uint64_t final_T;       /* W*H f32 */
    uint64_t n_contrib;     /* W*H u32 */
    uint64_t ranges;        /* tiles x uint2 */
    uint64_t qcount;        /* tiles x 4 u32: entries of each quadrant list */
} gs2m_layout;

int gs2m_debug_layout(int P, int R, int width, int height, gs2m_layout* out);

/* Binning mode.  Default (0): each Gaussian's tile rectangle is the reference's radius rectangle
 * intersected with the tiles its alpha >= 1/255 ellipse can reach; the dropped tiles cannot
 * contribute to any pixel, so every output is identical while fewer instances are processed
 * (num_rendered and the private tile lists shrink).  1: emit exactly the reference's rectangle
 * (cuda_rasterizer/auxiliary.h:44-53), giving bit-identical sorted lists / ranges / num_rendered;
 * used by the parity tests of the integer artefacts. */
int gs2m_set_reference_binning(int on);

/* ---- SH coefficients in two tensors ------------------------------------------------------------------------------
 * The reference model stores them that way (scene/gaussian_model.py: _features_dc (P,1,3), _features_rest (P,15,3))
 * and concatenates them for every view (get_features, 192 B per Gaussian forward and again in the backward).  These
 * two entry points take -- and in the backward return -- the two parts directly; everything else is
 * gs2m_raster_forward / gs2m_raster_backward.  M must be 16 (sh_rest holds M - 1 = 15 coefficients), both pointers
 * 16-byte aligned; otherwise GS2M_ERR_UNSUPPORTED (concatenate and use the plain entry points). */
int gs2m_raster_forward_split_sh(gs2m_alloc_fn geometry_alloc, void* geometry_user, gs2m_alloc_fn binning_alloc,
                        void* binning_user, gs2m_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D, const float* sh_dc, const float* sh_rest,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* features,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                        float tan_fovy, int prefiltered, int feature_count, float* out_color, int* out_radii,
                        int* out_observe, float* out_buffer, void* stream_);
int gs2m_raster_backward_split_sh(int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* sh_dc, const float* sh_rest, const float* colors_precomp, const float* scales,
                         float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* features, const float* viewmatrix, const float* projmatrix, const float* campos,
                         float tan_fovx, float tan_fovy, const int* radii, const float* buffer, char* geom_buffer,
                         char* binning_buffer, char* image_buffer, int feature_count, const float* grad_colors,
                         const float* grad_buffer, float* dL_dmeans2D, float* dL_dconics, float* dL_dopacities,
                         float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscales,
                         float* dL_drots, float* dL_dfeatures, gs2m_alloc_fn scratch_alloc, void* scratch_user,
                         void* stream_);

/* ---- render() pre/post-processing around the rasterizer ------------------------------
 * Replace ~25 small PyTorch ops per view of the reference's render() (three (N,3)x(3,3) matmuls and a bmm among
 * them) by four elementwise kernels.  All pointers are device pointers; `view` is world_view_transform, row-major
 * 4x4 with the reference's row-vector convention (p_cam = p @ view[:3,:3] + view[3,:3]).
 *
 * gs2m_pack_features_*: gaussian_renderer/__init__.py:83-96 + scene/gaussian_model.py:146-160
 *   (pc.get_normals(camera_center): min-scale axis of build_rotation(rotations), flipped to face the camera,
 *   normalised) -> features (P,10) = [1, distance, normal(3), albedo(3), roughness, metallic | 0], distance =
 *   |n_cam . p_cam| or p_cam.z when z_depth.  `scales`, `rotations`, `albedo`, ... are the ACTIVATED values
 *   (pc.get_scaling, pc.get_rotation, ...).  The backward returns the gradients autograd would: xyz, rotations,
 *   albedo, roughness, metallic (scales only select the axis: no gradient). */
int gs2m_pack_features_forward(int P, const float* xyz, const float* scales, const float* rotations, const float* albedo,
                               const float* roughness, const float* metallic, const float* campos, const float* view,
                               int z_depth, int blend_metallic, float* features, void* stream);
int gs2m_pack_features_backward(int P, const float* xyz, const float* scales, const float* rotations, const float* campos,
                                const float* view, int z_depth, int blend_metallic, const float* dL_dfeatures,
                                float* dL_dxyz, float* dL_drotations, float* dL_dalbedo, float* dL_droughness,
                                float* dL_dmetallic, void* stream);
/* gs2m_gbuffer_post_*: gaussian_renderer/__init__.py:126-141.  buffer (10,H,W) -> normal_mask (H*W bytes: all three
 * normal channels != 0), local_normal_map (3,H,W) = normal_map rotated into the camera frame, depth_map (1,H,W) =
 * distance / -(local_normal . ray + 1e-8) (or the distance channel itself when z_depth); rays (H*W,3) =
 * viewpoint_camera.get_rays(), unused when z_depth.  The backward writes channels 1..4 of dL_dbuffer (10,H,W);
 * either upstream gradient may be NULL (= zero). */
int gs2m_gbuffer_post_forward(int width, int height, const float* buffer, const float* rays, const float* view, int z_depth,
                              uint8_t* normal_mask, float* local_normal_map, float* depth_map, void* stream);
int gs2m_gbuffer_post_backward(int width, int height, const float* buffer, const float* rays, const float* view,
                               int z_depth, const float* dL_dlocal_normal, const float* dL_ddepth, float* dL_dbuffer,
                               void* stream);

/* The same backward with the gradients of the maps render() hands out as channel slices of the buffer (alpha 0,
 * distance 1, normal 2..4, albedo 5..7, roughness 8, metallic 9; each (k,H,W), NULL = none) folded in: writes ALL ten
 * channels of dL_dbuffer once (PyTorch's slicing backward zero-fills and adds a full (10,H,W) tensor per slice). */
int gs2m_gbuffer_maps_backward(int width, int height, const float* buffer, const float* rays, const float* view,
                               int z_depth, const float* dL_dlocal_normal, const float* dL_ddepth, const float* dL_dalpha,
                               const float* dL_ddistance, const float* dL_dnormal, const float* dL_dalbedo,
                               const float* dL_droughness, const float* dL_dmetallic, float* dL_dbuffer, void* stream);

/* gs2m_sobel_normal_*: render_normal_from_depth_map, gaussian_renderer/__init__.py:167-175 with
 * utils/normal_utils.py:3-72 (depth -> world points -> cross product of the central differences -> normalize, 0 on the
 * border; blended with the background by alpha).  depth, alpha: (H,W); bg: 3 floats; view: world_view_transform;
 * fx, fy, cx, cy: the zero-skew intrinsics of get_calib_matrix_nerf(); sobel_map: (3,H,W).  The backward returns
 * dL/ddepth and dL/dalpha (gather form, no atomics). */
int gs2m_sobel_normal_forward(int width, int height, const float* depth, const float* alpha, const float* bg,
                              const float* view, float fx, float fy, float cx, float cy, float* sobel_map, void* stream);
int gs2m_sobel_normal_backward(int width, int height, const float* depth, const float* alpha, const float* bg,
                               const float* view, float fx, float fy, float cx, float cy, const float* dL_dsobel,
                               float* dL_ddepth, float* dL_dalpha, void* stream);

/* The GaussianModel getters render() calls for every view (scene/gaussian_model.py:113-144): scales = exp(_scaling),
 * rotations = normalize(_rotation) (torch.nn.functional.normalize: x / max(|x|, 1e-12)), opacity / albedo / roughness /
 * metallic = sigmoid(raw), as ONE launch forward and ONE backward instead of ~9 + ~14 PyTorch launches.  Any input
 * pointer may be NULL (that activation is skipped; its output pointer is ignored).  Backward: a NULL dL_d<raw> output
 * skips that group (used when the loss does not reach it, e.g. metallic without blend_metallic); the activated values
 * are the forward's outputs, `rotation` is the raw quaternion.  (P,4) arrays must be 16-byte aligned. */
int gs2m_activate_forward(int P, const float* scaling, const float* rotation, const float* opacity, const float* albedo,
                          const float* roughness, const float* metallic, float* scales, float* rotations, float* opacities,
                          float* albedo_a, float* roughness_a, float* metallic_a, void* stream);
int gs2m_activate_backward(int P, const float* rotation, const float* scales, const float* opacities, const float* albedo_a,
                           const float* roughness_a, const float* metallic_a, const float* dL_dscales,
                           const float* dL_drotations, const float* dL_dopacities, const float* dL_dalbedo_a,
                           const float* dL_droughness_a, const float* dL_dmetallic_a, float* dL_dscaling, float* dL_drotation,
                           float* dL_dopacity, float* dL_dalbedo, float* dL_droughness, float* dL_dmetallic, void* stream);

/* Forward: how the host waits for num_rendered.  1 (default): it polls the pinned landing zone of the 4-byte copy
 * (falls back to a stream synchronize after 2 s); 0: hipStreamSynchronize.  Same results. */
int gs2m_set_spin_wait(int on);

/* 1: the tile sort hands its 4096-key tiles out by an atomic ticket instead of by workgroup id, so that a tile only ever waits for
 * tiles that have started -- needed when other kernels may be resident on the device at the same time (RCCL's in data-parallel
 * runs: gs2m_dp.GradReducer switches it on; a CU-masked or partitioned device).  0 (default): workgroup ids whenever every
 * workgroup of a pass fits on the device at once.  Also: environment variable GS2M_SORT_TICKETS=1. */
int gs2m_set_sort_tickets(int on);

/* Which kernels put a tile's span into (depth, id) order (csrc/tile_sort.hip).  0 (default): by the number of tiles -- a frame of at
 * least 2560 tiles gives every tile one wave (spans of up to 1024 entries), a smaller frame gives every tile a workgroup; 1: workgroups
 * whatever the frame; 2: waves whatever the frame; 3: waves, with the spans of 513 .. 1024 entries left to the workgroup kernel that
 * takes the longer ones (what 0 and 2 choose by themselves when a frame's average span is at most 400 entries).  Spans of more than
 * 1024 entries go to a workgroup in every setting.  Same results. */
int gs2m_set_tile_sort_policy(int policy);

/* A ready-made gs2m_alloc_fn for callers that want the binning buffer (sized only after the forward's one host wait)
 * allocated AHEAD of that wait: pass gs2m_prealloc_alloc as the callback and a gs2m_prealloc as its user pointer.  A
 * request that fits `capacity` returns `ptr` without leaving the library -- the GPU idles between the wait and the next
 * launch, so a Python allocator call there costs a few per cent of a step on a slow host; a larger request goes to
 * `fallback` (and sets used_fallback). */
typedef struct gs2m_prealloc {
    char* ptr;
    size_t capacity;
    gs2m_alloc_fn fallback;
    void* fallback_user;
    int used_fallback;
    size_t requested; /* out: bytes of the last request (a caller that keeps `ptr` from call to call sizes / shrinks it by this) */
} gs2m_prealloc;
char* gs2m_prealloc_alloc(size_t bytes, void* user);

/* Debug mode ("race detection / sanitizers"): 1 = after every pipeline stage the stream is
 * synchronized and checked; a fault is returned as GS2M_ERR_STAGE(stage) by the call that launched it, with
 * gs2m_stage_name(stage) naming it.  0 (default): launch errors only, checked once per call.
 * Markers: 1 = roctx ranges named after the stages around their launches (rocprofv3 --marker-trace); needs
 * libroctx64.so at run time (GS2M_ERR_UNSUPPORTED otherwise). */
int gs2m_set_debug(int on);
int gs2m_set_markers(int on);
const char* gs2m_stage_name(int stage);

/* Limits: num_rendered (the emitted instance count) must stay below 2^29 -- slots and list offsets are 32-bit, gradient rows (up to 4
 * per instance, 256 per unit of a heavy Gaussian) stay below 2^31 --; P below 2^28 (a 4-bit quadrant mask rides above the Gaussian id);
 * at most 2^28 tiles, 2^22 heavy units.
 * gs2m_raster_forward returns GS2M_ERR_UNSUPPORTED beyond that. */

/* ---- backward scratch sized by what the forward actually binned -------------------------------------------------------
 * The blend backward writes one partial-gradient row (11 + feature_count floats, padded to a multiple of 4) per
 * (tile instance, 8x8 quadrant the splat reaches); the rows are numbered densely, and their number -- 1.8 per instance on
 * the bench scene, 4 at most -- is known on the device shortly after the forward's binning.  A caller that wants the scratch
 * sized by it (instead of the worst case 4 x num_rendered):
 *   token = gs2m_raster_forward_token()       right after a forward, on the calling thread (no wait);
 *   rows  = gs2m_raster_dense_rows(token)     any time later, from any thread (waits for the count if the GPU has not
 *                                             produced it yet; -1 when it is no longer available: size for the worst case);
 *   gs2m_raster_backward_rows_hint(rows)      on the thread that calls gs2m_raster_backward* next (consumed by that call). */
unsigned long long gs2m_raster_forward_token(void);
long long gs2m_raster_dense_rows(unsigned long long token);
int gs2m_raster_backward_rows_hint(long long dense_rows);

/* Test hook (tests/test_tile_sort_gpu.py): the per-tile (depth, id) sort + quadrant-list split on caller-made spans. */
int gs2m_debug_tile_sort(int tiles, const unsigned* ranges_raw, unsigned* ranges, const unsigned* slot_sorted, const unsigned* e_rec,
                         const unsigned* wave_rowbase, unsigned* point_list, unsigned* row_tmp,
                         unsigned* qlist, unsigned* qrow, unsigned* qcount, void* stream);

/* ---- per-stage timing with HIP events recorded on the launch stream (bench.py) ----
 * mode 0 = off, 1 = the two blend kernels only, 2 = every stage, 3 = the backward blend kernel only.  Setting the mode clears
 * the records.  gs2m_profile_collect waits for the recorded events and returns, per
 * stage, the summed milliseconds and the number of launches since the last collect.
 * Stage order: preprocess, -, scan (num_rendered + block prefixes), emit (+ row scan), tile_sort (stable radix sort by tile),
 * depth_order+quad_lists (per-tile sort on chip, ranges, quadrant lists), blend_fwd, -, blend_bwd, gaussian_bwd (row sums +
 * per-Gaussian chain). */
#define GS2M_NUM_STAGES 10
int gs2m_profile_mode(int mode);
/* mode 3 brackets every `every`-th launch of the backward blend (default 1: each one).  An event pair leaves ~6 us of bubble on
 * the stream on either side of the kernel it brackets; a caller that times a loop around the launches it measures (bench.py)
 * samples them instead. */
int gs2m_profile_sampling(int every);
int gs2m_profile_collect(float* stage_ms, int* stage_count, int n_stages);

/* Library version / build info string (static storage). */
const char* gs2m_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GS2M_RASTER_H */
