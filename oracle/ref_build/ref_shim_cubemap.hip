// TEST INFRASTRUCTURE ONLY: a C ABI in front of the reference's environment-map prefilter kernels
// (submodules/render-utils/c_src/cubemap.cu, translated by hipify-perl at build time: oracle/ref_build/Makefile).  This file
// is this repository's code.  It does what the reference's torch binding does around each launch
// (c_src/torch_bindings.cpp:740-889: grid = the cube map's (res, res, 6), launch sizes from the reference's own
// getLaunchBlockSize / getLaunchGridSize with its 8 x 8 block, `Tensor` descriptors of contiguous fp32 tensors as
// make_cuda_tensor fills them, :130-156) with plain device pointers instead of torch tensors.  Outputs that the binding
// creates with torch::zeros must arrive zeroed.
#include <hip/hip_runtime.h>

#include <cstring>

#include "cubemap.h"

__global__ void DiffuseCubemapFwdKernel(DiffuseCubemapKernelParams p);
__global__ void DiffuseCubemapBwdKernel(DiffuseCubemapKernelParams p);
__global__ void SpecularBoundsKernel(SpecularBoundsKernelParams p);
__global__ void SpecularCubemapFwdKernel(SpecularCubemapKernelParams p);
__global__ void SpecularCubemapBwdKernel(SpecularCubemapKernelParams p);

namespace {

Tensor describe(const void* val, int res, int channels, dim3 out_dims) {
    Tensor t;
    std::memset(&t, 0, sizeof(t));
    const int dims[4] = {6, res, res, channels};
    int stride = 1;
    for (int i = 3; i >= 0; i--) {
        t.dims[i] = dims[i];
        t.strides[i] = stride;
        stride *= dims[i];
    }
    t._dims[0] = (int)out_dims.z; t._dims[1] = (int)out_dims.y; t._dims[2] = (int)out_dims.x; t._dims[3] = channels;
    t.fp16 = false;
    t.val = const_cast<void*>(val);
    t.d_val = nullptr;
    return t;
}

int done() { return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -1; }

struct Launch {
    dim3 grid_size, block, grid;
    explicit Launch(int res) : grid_size(res, res, 6) {
        block = getLaunchBlockSize(8, 8, grid_size);
        grid = getLaunchGridSize(block, grid_size);
    }
};

}  // namespace

extern "C" {

// cubemap (6, res, res, 3) -> out (6, res, res, 3)
int gs2m_ref_diffuse_cubemap_fwd(int res, const float* cubemap, float* out) {
    Launch L(res);
    DiffuseCubemapKernelParams p;
    p.gridSize = L.grid_size;
    p.cubemap = describe(cubemap, res, 3, L.grid_size);
    p.out = describe(out, res, 3, L.grid_size);
    DiffuseCubemapFwdKernel<<<L.grid, L.block>>>(p);
    return done();
}

// grad (6, res, res, 3) -> cubemap_grad (6, res, res, 3), ZEROED by the caller (the kernel accumulates with atomicAdd)
int gs2m_ref_diffuse_cubemap_bwd(int res, const float* cubemap, const float* grad, float* cubemap_grad) {
    Launch L(res);
    DiffuseCubemapKernelParams p;
    p.gridSize = L.grid_size;
    p.cubemap = describe(cubemap, res, 3, L.grid_size);
    p.out = describe(grad, res, 3, L.grid_size);
    p.cubemap.d_val = cubemap_grad;
    DiffuseCubemapBwdKernel<<<L.grid, L.block>>>(p);
    return done();
}

// -> out (6, res, res, 24), ZEROED by the caller
int gs2m_ref_specular_bounds(int res, float costheta_cutoff, float* out) {
    Launch L(res);
    SpecularBoundsKernelParams p;
    p.costheta_cutoff = costheta_cutoff;
    p.gridSize = L.grid_size;
    p.out = describe(out, res, 24, L.grid_size);
    SpecularBoundsKernel<<<L.grid, L.block>>>(p);
    return done();
}

// cubemap (6, res, res, 3), bounds (6, res, res, 24) -> out (6, res, res, 4): weighted colour sums and the weight sum
int gs2m_ref_specular_cubemap_fwd(int res, const float* cubemap, const float* bounds, float roughness, float costheta_cutoff, float* out) {
    Launch L(res);
    SpecularCubemapKernelParams p;
    p.roughness = roughness;
    p.costheta_cutoff = costheta_cutoff;
    p.gridSize = L.grid_size;
    p.cubemap = describe(cubemap, res, 3, L.grid_size);
    p.bounds = describe(bounds, res, 24, L.grid_size);
    p.out = describe(out, res, 4, L.grid_size);
    SpecularCubemapFwdKernel<<<L.grid, L.block>>>(p);
    return done();
}

// grad (6, res, res, 4) -> cubemap_grad (6, res, res, 3), ZEROED by the caller
int gs2m_ref_specular_cubemap_bwd(int res, const float* cubemap, const float* bounds, const float* grad, float roughness,
                                  float costheta_cutoff, float* cubemap_grad) {
    Launch L(res);
    SpecularCubemapKernelParams p;
    p.roughness = roughness;
    p.costheta_cutoff = costheta_cutoff;
    p.gridSize = L.grid_size;
    p.cubemap = describe(cubemap, res, 3, L.grid_size);
    p.bounds = describe(bounds, res, 24, L.grid_size);
    p.out = describe(grad, res, 4, L.grid_size);
    p.cubemap.d_val = cubemap_grad;
    SpecularCubemapBwdKernel<<<L.grid, L.block>>>(p);
    return done();
}

}  // extern "C"
