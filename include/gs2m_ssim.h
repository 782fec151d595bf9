/* gs2m_ssim.h -- C ABI of the fused SSIM map, part of libgs2m_raster.so.
 *
 * Replaces the reference's CUDA-only dependency submodules/fused-ssim (bound at fused_ssim/__init__.py:4 as
 * `fused_ssim_cuda.fusedssim` / `fusedssim_backward`, declared in submodules/fused-ssim/ssim.h:7-26), which
 * train.py:103 and :136 call for the D-SSIM loss term; the operator is utils/loss_utils.py:30-70 `ssim`: an 11x11
 * separable Gaussian window (sigma 1.5) with zero "same" padding applied per channel, C1 = 0.01^2, C2 = 0.03^2.
 *
 * All pointers are DEVICE pointers to contiguous fp32 (B, CH, H, W) arrays.  No torch types; the Python mirror is
 * the package gs-2m_amd/fused_ssim (same `fused_ssim(img1, img2, padding="same", train=True)` and `FusedSSIMMap`).
 * Calls are asynchronous on `stream`; return GS2M_OK (0) or a negative GS2M_ERR_* code (gs2m_raster.h). */
#ifndef GS2M_SSIM_H
#define GS2M_SSIM_H

#ifdef __cplusplus
extern "C" {
#endif

/* fusedssim(C1, C2, img1, img2, train), ssim.h:7-14.  Writes ssim_map and, when the three derivative pointers are
 * non-NULL (train = True; all three or none), d map / d mu1, d map / d sigma1^2, d map / d sigma12 per element. */
int gs2m_ssim_forward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                      float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);

/* fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12), ssim.h:16-26 (C1, C2 are
 * not needed once the derivative maps exist).  Writes dL/dimg1; img2 is the ground truth and gets no gradient. */
int gs2m_ssim_backward(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                       const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1,
                       void* stream);

/* The same with a map gradient that is uniform: dL/dmap = dL_dvalue[0] * mul / div at every element (dL_dvalue: a
 * DEVICE scalar).  This is the backward of `ssim_map.mean()` (mul = 1, div = element count: fused_ssim/__init__.py:40-41)
 * and of the D-SSIM loss term lambda * (1 - mean) (mul = -lambda; train.py:103) without materialising the map's gradient. */
int gs2m_ssim_backward_uniform(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dvalue,
                               float mul, float div, const float* dm_dmu1, const float* dm_dsigma1_sq,
                               const float* dm_dsigma12, float* dL_dimg1, void* stream);

#ifdef __cplusplus
}
#endif
#endif
