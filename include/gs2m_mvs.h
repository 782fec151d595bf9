/* gs2m_mvs.h -- C ABI of the photometric multi-view term, part of libgs2m_raster.so.
 *
 * The plane-induced patch warp + normalised cross-correlation of multi_view_loss (utils/loss_utils.py:303-349, with
 * _patch_offsets :451-454, _patch_warp :456-466, _loss_ncc :468-509) as one kernel each way, for N sampled pixels:
 *     H_i  = M - b (n_i^T Kinv) / d_i                   (M = K_near R_rn K_ref^-1, b = K_near t_rn: HOST 3x3 / 3 arrays, row major)
 *     ncc_i = clamp(1 - cross^2 / (var_ref var_near + 1e-8), 0, 2)  over the (2 patch + 1)^2 positions around pixels_i / ncc_scale,
 * both grey images (height, width) sampled bilinearly with zero padding (F.grid_sample, align_corners=True).
 * Backward: gradients to the normals (N, 3) and distances (N) only, as in the reference (everything else is detached or data).
 * Device pointers, fp32.  Asynchronous on `stream`; return GS2M_OK (0) or a negative GS2M_ERR_* code (gs2m_raster.h). */
#ifndef GS2M_MVS_H
#define GS2M_MVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int gs2m_patch_ncc_forward(int N, const float* pixels, const float* normals, const float* dists, const float* ref_gray,
                           const float* near_gray, int width, int height, const float* M, const float* b, const float* Kinv,
                           float ncc_scale, int patch, float* ncc, void* stream);
int gs2m_patch_ncc_backward(int N, const float* pixels, const float* normals, const float* dists, const float* ref_gray,
                            const float* near_gray, int width, int height, const float* M, const float* b, const float* Kinv,
                            float ncc_scale, int patch, const float* dL_dncc, float* dL_dnormals, float* dL_ddists, void* stream);

/* roughness_loss (utils/loss_utils.py:138-243), forward only as there: the grey-value NCC, the NCC of the 3x3-Sobel gradient
 * magnitudes of the two patches (_patch_gradient :232-238) and the reference patch's unnormalised variance (the low-texture
 * switch sqrt(ref_var) < 0.01, :209-211), each (N).  patch <= 3. */
int gs2m_patch_ncc_roughness(int N, const float* pixels, const float* normals, const float* dists, const float* ref_gray,
                             const float* near_gray, int width, int height, const float* M, const float* b, const float* Kinv,
                             float ncc_scale, int patch, float* ncc_gray, float* ncc_grad, float* ref_var, void* stream);

/* The two backwards below scatter bilinear footprints into the neighbour's maps.  Deterministic mode (default ON): the sums are
 * formed in 64-bit fixed point scaled by the call's largest contribution (integer atomics: independent of the order of the
 * additions, bitwise reproducible run to run) in a per-(device, stream) workspace of the library's own; off: fp32 atomics (the
 * last bits of a texel then depend on the order in which the hardware retires them).  Process-wide. */
void gs2m_mvs_set_deterministic(int on);
int gs2m_mvs_get_deterministic(void);

/* torch.nn.functional.grid_sample(image[None], grid.view(1, N, 1, 2), mode='bilinear', padding_mode='border',
 * align_corners=True) for a (channels, height, width) image at N normalised positions -- how _sample_depth_normal
 * (utils/loss_utils.py:368-409) looks the neighbour's depth and normal maps up; out, dL_dout: (N, channels), channels <= 4.
 * Backward: dL_dimage (same shape as image) is ACCUMULATED into (zero it first), dL_dgrid (N, 2) is written; either may be NULL. */
int gs2m_grid_sample_border_forward(int N, int channels, int height, int width, const float* image, const float* grid, float* out,
                                    void* stream);
int gs2m_grid_sample_border_backward(int N, int channels, int height, int width, const float* image, const float* grid,
                                     const float* dL_dout, float* dL_dimage, float* dL_dgrid, void* stream);

/* The geometric half of multi_view_loss per pixel (utils/loss_utils.py:256-283): pixel -> reference-camera point (rendered depth)
 * -> neighbour camera (Y = A P + b) -> projection -> bilinear border lookup of the neighbour's depth / normal maps -> reprojection
 * with that depth (Z = A2 Y' + b2) -> pixel_noise = distance to the pixel; angle = acos(clamp(n_ref . n_near)); valid = projects
 * inside, z > 0.1, passes the occlusion test.  depth (H, W), normal (3, H, W), neighbour maps (Hn, Wn) / (3, Hn, Wn); A, A2 (3x3 row
 * major), b, b2 (3), intr_* = (fx, fy, cx, cy): HOST arrays.  Backward: dL_ddepth (H, W) and dL_dnormal (3, H, W) are written,
 * dL_ddepth_n / dL_dnormal_n are ACCUMULATED into (zero them first). */
int gs2m_mv_geo_forward(int width, int height, int width_n, int height_n, const float* depth, const float* normal, const float* depth_n,
                        const float* normal_n, const float* A, const float* b, const float* A2, const float* b2, const float* intr_ref,
                        const float* intr_near, float occlusion, float* pixel_noise, float* angle, uint8_t* valid, void* stream);
int gs2m_mv_geo_backward(int width, int height, int width_n, int height_n, const float* depth, const float* normal, const float* depth_n,
                         const float* normal_n, const float* A, const float* b, const float* A2, const float* b2, const float* intr_ref,
                         const float* intr_near, float occlusion, const float* dL_dnoise, const float* dL_dangle, float* dL_ddepth,
                         float* dL_dnormal, float* dL_ddepth_n, float* dL_dnormal_n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
