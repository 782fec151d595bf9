// TEST INFRASTRUCTURE ONLY (like everything under oracle/): a C ABI in front of the REFERENCE's own rasterizer
// (CudaRasterizer::Rasterizer::forward / backward, cuda_rasterizer/rasterizer.h:25-90), whose CUDA sources are passed
// through the image's hipify-perl at build time by oracle/ref_build/Makefile and compiled for gfx950 into
// oracle/_ref/libgs2m_ref.so.  This file is this repository's code: it holds the three state buffers the reference asks
// its caller to allocate (rasterize_points.cu:35-43 does it with torch tensors), forwards every argument unchanged and
// copies the reference's internal arrays out for the parity tests.  All data pointers are DEVICE pointers; everything
// runs on the null stream, as the reference does, and returns after a device synchronisation.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <functional>

#include "rasterizer.h"
#include "rasterizer_impl.h"
#include "simple_knn.h"

namespace {

struct Chunk {
    char* p = nullptr;
    size_t cap = 0;
    char* need(size_t n) {
        if (n > cap) {
            if (p) (void)hipFree(p);
            p = nullptr;
            cap = 0;
            if (hipMalloc(reinterpret_cast<void**>(&p), n) != hipSuccess) return nullptr;
            cap = n;
        }
        return p;
    }
    ~Chunk() {
        if (p) (void)hipFree(p);
    }
};

struct Handle {
    Chunk geom, binning, image;
    int P = 0, R = 0, N = 0;
};

int sync_status() { return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -1; }

template <typename T>
int copy_out(void* dst, const T* src, size_t count) {
    if (!dst) return 0;
    return hipMemcpy(dst, src, count * sizeof(T), hipMemcpyDeviceToDevice) == hipSuccess ? 0 : -1;
}

}  // namespace

extern "C" {

void* gs2m_ref_create(void) { return new Handle(); }

void gs2m_ref_destroy(void* h) { delete static_cast<Handle*>(h); }

// -> num_rendered (>= 0), or -1
int gs2m_ref_forward(void* handle, int P, int D, int M, const float* background, int width, int height, const float* means3D,
                     const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                     float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* features,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     int prefiltered, int featureCount, float* out_colors, int* out_radii, int* out_observe, float* out_buffer) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return -1;
    int R = -1;
    try {
        R = CudaRasterizer::Rasterizer::forward(
            [h](size_t n) { return h->geom.need(n); }, [h](size_t n) { return h->binning.need(n); },
            [h](size_t n) { return h->image.need(n); }, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities,
            scales, scale_modifier, rotations, cov3D_precomp, features, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
            prefiltered != 0, featureCount, out_colors, out_radii, out_observe, out_buffer);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "gs2m_ref_forward: %s\n", e.what());
        return -1;
    }
    if (sync_status() != 0) return -1;
    h->P = P;
    h->R = R;
    h->N = width * height;
    return R;
}

int gs2m_ref_backward(void* handle, int P, int D, int M, int R, const float* background, int width, int height,
                      const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* features,
                      const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                      const int* radii, const float* buffer, int featureCount, const float* grad_colors, const float* grad_buffer,
                      float* dL_dmeans2D, float* dL_dconics, float* dL_dopacities, float* dL_dcolors, float* dL_dmeans3D,
                      float* dL_dcov3D, float* dL_dshs, float* dL_dscales, float* dL_drots, float* dL_dfeatures) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !h->geom.p || !h->image.p || (R > 0 && !h->binning.p)) return -1;
    try {
        CudaRasterizer::Rasterizer::backward(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier,
                                             rotations, cov3D_precomp, features, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                                             buffer, h->geom.p, h->binning.p, h->image.p, featureCount, grad_colors, grad_buffer,
                                             dL_dmeans2D, dL_dconics, dL_dopacities, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dshs, dL_dscales,
                                             dL_drots, dL_dfeatures);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "gs2m_ref_backward: %s\n", e.what());
        return -1;
    }
    return sync_status();
}

// The reference's internal arrays of the last forward (any destination may be NULL): geometry state (P entries), the
// sorted instance list (R), the per-pixel state (N = width x height).
int gs2m_ref_state(void* handle, float* z_depths, int* internal_radii, float* means2D /* 2 P */, float* cov3D /* 6 P */,
                   float* conic_opacity /* 4 P */, float* rgb /* 3 P */, uint32_t* tiles_touched, uint32_t* point_offsets,
                   uint8_t* clamped /* 3 P */, uint64_t* point_list_keys, uint32_t* point_list, uint32_t* ranges /* 2 N (first tiles used) */,
                   uint32_t* n_contrib, float* accum_alpha) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !h->geom.p) return -1;
    const size_t P = (size_t)h->P, R = (size_t)h->R, N = (size_t)h->N;
    char* c = h->geom.p;
    const CudaRasterizer::GeometryState g = CudaRasterizer::GeometryState::fromChunk(c, P);
    int rc = 0;
    rc |= copy_out(z_depths, g.z_depths, P);
    rc |= copy_out(internal_radii, g.internal_radii, P);
    rc |= copy_out(means2D, reinterpret_cast<const float*>(g.means2D), 2 * P);
    rc |= copy_out(cov3D, g.cov3D, 6 * P);
    rc |= copy_out(conic_opacity, reinterpret_cast<const float*>(g.conic_opacity), 4 * P);
    rc |= copy_out(rgb, g.rgb, 3 * P);
    rc |= copy_out(tiles_touched, g.tiles_touched, P);
    rc |= copy_out(point_offsets, g.point_offsets, P);
    rc |= copy_out(clamped, reinterpret_cast<const uint8_t*>(g.clamped), 3 * P);
    if (R > 0 && h->binning.p) {
        char* b = h->binning.p;
        const CudaRasterizer::BinningState s = CudaRasterizer::BinningState::fromChunk(b, R);
        rc |= copy_out(point_list_keys, s.point_list_keys, R);
        rc |= copy_out(point_list, s.point_list, R);
    }
    char* i = h->image.p;
    const CudaRasterizer::ImageState im = CudaRasterizer::ImageState::fromChunk(i, N);
    rc |= copy_out(ranges, reinterpret_cast<const uint32_t*>(im.ranges), 2 * N);
    rc |= copy_out(n_contrib, im.n_contrib, N);
    rc |= copy_out(accum_alpha, im.accum_alpha, N);
    if (hipDeviceSynchronize() != hipSuccess) rc = -1;
    return rc;
}

// markVisible (rasterizer.h:19-24); present: P bytes
int gs2m_ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, reinterpret_cast<bool*>(present));
    return sync_status();
}

// distCUDA2's kernel side (simple-knn/spatial.cu:15-26 calls SimpleKNN::knn on the (P, 3) points): mean squared distance to
// the three nearest neighbours
int gs2m_ref_knn(int P, const float* points, float* mean_dists) {
    SimpleKNN::knn(P, reinterpret_cast<float3*>(const_cast<float*>(points)), mean_dists);
    return sync_status();
}

}  // extern "C"
