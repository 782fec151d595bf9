// TEST INFRASTRUCTURE ONLY: a C ABI in front of the reference's fused-ssim extension (submodules/fused-ssim/ssim.cu, translated
// by hipify-perl at build time: oracle/ref_build/Makefile).  This file is this repository's code.  The extension's two host
// functions take and return torch tensors (ssim.h); they are called as they are, on tensors that alias the caller's device
// buffers (torch::from_blob), and their results are copied into the caller's buffers.  Library: oracle/_ref/libgs2m_ref_ssim.so
// (links libtorch: loaded from a process that has imported torch).
#include <hip/hip_runtime.h>
#include <torch/extension.h>

#include "ssim.h"

namespace {

torch::Tensor alias(const float* p, int B, int CH, int H, int W) {
    return torch::from_blob(const_cast<float*>(p), {B, CH, H, W}, torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA));
}

int copy_out(float* dst, const torch::Tensor& t) {
    if (!dst || t.numel() == 0) return 0;
    return hipMemcpy(dst, t.data_ptr<float>(), (size_t)t.numel() * sizeof(float), hipMemcpyDeviceToDevice) == hipSuccess ? 0 : -1;
}

}  // namespace

extern "C" {

// fusedssim(C1, C2, img1, img2, train) (ssim.cu:368-404): train = (dm_dmu1 != NULL)
int gs2m_ref_ssim_forward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, float* ssim_map,
                          float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12) {
    try {
        torch::Tensor a = alias(img1, B, CH, H, W), b = alias(img2, B, CH, H, W);
        auto r = fusedssim(C1, C2, a, b, dm_dmu1 != nullptr);
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        int rc = copy_out(ssim_map, std::get<0>(r));
        rc |= copy_out(dm_dmu1, std::get<1>(r)) | copy_out(dm_dsigma1_sq, std::get<2>(r)) | copy_out(dm_dsigma12, std::get<3>(r));
        return hipDeviceSynchronize() == hipSuccess ? rc : -1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "gs2m_ref_ssim_forward: %s\n", e.what());
        return -1;
    }
}

// fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12) (ssim.cu:406-443)
int gs2m_ref_ssim_backward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, const float* dL_dmap,
                           const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1) {
    try {
        torch::Tensor a = alias(img1, B, CH, H, W), b = alias(img2, B, CH, H, W), g = alias(dL_dmap, B, CH, H, W);
        torch::Tensor m = alias(dm_dmu1, B, CH, H, W), s1 = alias(dm_dsigma1_sq, B, CH, H, W), s12 = alias(dm_dsigma12, B, CH, H, W);
        torch::Tensor r = fusedssim_backward(C1, C2, a, b, g, m, s1, s12);
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        const int rc = copy_out(dL_dimg1, r);
        return hipDeviceSynchronize() == hipSuccess ? rc : -1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "gs2m_ref_ssim_backward: %s\n", e.what());
        return -1;
    }
}

}  // extern "C"
