/* gs2m_loss.h -- C ABI of the fused loss tail of one training iteration ("optimizer-side
 * elementwise work"), part of libgs2m_raster.so.
 *
 * The reference computes these terms with plain PyTorch ops (there is no native interface to replace); each entry
 * point below cites the Python it restates.  The Python mirror is gs-2m_amd/gs2m_losses.py (`geometry_image_loss`,
 * `fused_plane_loss`, `edge_gradient`) and GaussianModel.add_densification_stats / update_max_radii.
 *
 * All pointers are DEVICE pointers to contiguous arrays (fp32 unless stated); images are planar (C, H, W).  Scalars that
 * the training loop produces on the device (upstream gradients, min / max) are passed as device pointers so that no
 * call synchronises.  `workspace`: gs2m_loss_workspace_bytes() bytes of device memory, ZERO when first used and owned
 * by one stream at a time (the reducing kernels leave it zeroed).  Calls are asynchronous on `stream`; return GS2M_OK (0)
 * or a negative GS2M_ERR_* code (gs2m_raster.h).  Sums are formed in a fixed order: results are bitwise reproducible. */
#ifndef GS2M_LOSS_H
#define GS2M_LOSS_H

#ifdef __cplusplus
extern "C" {
#endif

int gs2m_loss_workspace_bytes(void);

/* _get_img_grad_weight, utils/loss_utils.py:122-135, up to its min-max normalisation: edge[y][x] = max over the two axes
 * of the channel-mean absolute central difference of gt (3, H, W) for interior pixels, 0 on the one-pixel border;
 * edge_minmax[0..1] = min and max of edge over the interior.  Depends on the ground truth only. */
int gs2m_edge_gradient(int W, int H, const float* gt, float* edge, float* edge_minmax, void* workspace, void* stream);

/* train.py:101-104, 113-120 (and :141-146 for the shaded image of the material stage) in one pass:
 *   rgb = clamp(image, 0, 1), or `background` (3) where mask (H, W bytes, torch.bool) is 0   (written, (3, H, W): the
 *         D-SSIM term reads it); image is (3, H, W), or (H, W, 3) with image_hwc != 0 (the layout pbr_shading returns),
 *   l1 = mean |rgb - gt|                                            (l1_loss, utils/loss_utils.py:27-28)
 *   dn = mean_pixels( w * sum_c |sobel_map - normal_map| ),  w = clamp(1 - (edge - min) / (max - min), 0, 1)^2 inside,
 *        1 on the border, times weight_map when given               (depth_normal_loss, utils/loss_utils.py:113-120)
 * out[0] = w_l1 * l1 + w_dn * dn, out[1] = l1, out[2] = dn.  normal_map / sobel_map both NULL: no dn term;
 * edge / edge_minmax both NULL: w = 1; weight_map may be NULL. */
int gs2m_image_loss_forward(int W, int H, const float* image, int image_hwc, const unsigned char* mask, const float* background,
                            const float* gt, const float* normal_map, const float* sobel_map,
                            const float* edge, const float* edge_minmax, const float* weight_map, float w_l1, float w_dn,
                            float* rgb, float* out, void* workspace, void* stream);

/* Backward of the above: g_loss[0] = d L / d out[0] (NULL: 0), g_rgb = d L / d rgb from the D-SSIM term (NULL: 0).
 * Writes d_image (in the image's layout; through the clamp and the mask: 0 outside [0, 1] and where mask is 0) and, when the
 * dn term is present, d_normal_map and d_sobel_map. */
int gs2m_image_loss_backward(int W, int H, const float* image, int image_hwc, const unsigned char* mask, const float* gt,
                             const float* normal_map, const float* sobel_map,
                             const float* edge, const float* edge_minmax, const float* weight_map, float w_l1, float w_dn,
                             const float* g_loss, const float* g_rgb, float* d_image, float* d_normal_map, float* d_sobel_map,
                             void* stream);

/* tv_loss, utils/loss_utils.py:536-557 (the smoothness terms of the material stage, train.py:160-175): edge-aware total
 * variation of pred (C, H, W) against the ground truth gt (3, H, W): neighbour differences along both image axes, absolute
 * (norm1 != 0) or squared, each damped by exp(-channel-mean |gt difference|) of the same pixel pair and, when weight_map
 * (H, W) is given, by the mean of the pair's two weights; out[0] = mean over the vertical pairs + mean over the horizontal
 * pairs, times `weight` (the term's lambda in the loss: no one-element multiply kernels around the node).  The backward
 * writes d_pred = g_loss[0] * d out / d pred (gt and weight_map get no gradient, as in the loop). */
int gs2m_tv_loss_forward(int W, int H, int C, const float* gt, const float* pred, const float* weight_map, int norm1, float weight,
                         float* out, void* workspace, void* stream);
int gs2m_tv_loss_backward(int W, int H, int C, const float* gt, const float* pred, const float* weight_map, int norm1, float weight,
                          const float* g_loss, float* d_pred, void* stream);

/* The geometric part of multi_view_loss, utils/loss_utils.py:277-291, from the per-pixel outputs of gs2m_mv_geo_forward
 * (gs2m_mvs.h): pixel_valid = valid & (noise < 1) and w_ncc = exp(-noise) on it (both written: the photometric part samples
 * from them), angle_valid = valid & (angle < angle_threshold), geo_w = exp(-decay * noise) on pixel_valid (a detached weight),
 * out[0] = weight * (sum geo_w noise / #pixel_valid + sum geo_w factor angle over angle_valid / #angle_valid) (empty sets
 * count 1), out[1..2] = the two counts.  valid / pixel_valid: bytes (torch.bool).  The backward writes d_noise, d_angle. */
int gs2m_mv_geo_loss_forward(int n, const float* noise, const float* angle, const unsigned char* valid, float angle_threshold, float decay,
                             float factor, float weight, float* out, unsigned char* pixel_valid, float* w_ncc, void* workspace, void* stream);
int gs2m_mv_geo_loss_backward(int n, const float* noise, const float* angle, const unsigned char* valid, float angle_threshold, float decay,
                              float factor, float weight, const float* out, const float* g_loss, float* d_noise, float* d_angle,
                              void* stream);

/* The photometric part of multi_view_loss around the patch NCC (utils/loss_utils.py:293-300, 345-349; gs-2m_amd/gs2m_mvs.py).
 * mv_take: for n DISTINCT pixel indices idx (int64, y * width + x) -> pixels (n,2) = (x, y), normals (n,3) from normal_map (3,H,W),
 * dists (n) from dist_map (H,W), w (n) from w_map (H,W; NULL: ones).  The backward writes d_normals / d_dists into d_normal_map (3,H,W) /
 * d_dist_map (H,W), which the caller has zero-filled.
 * ncc_tail: out[0] = sum(ncc w [ncc < 0.9]) / max(#[ncc < 0.9], 1), out[1] = that count; the backward writes d_ncc (w carries no gradient). */
int gs2m_mv_take_forward(int n, const long long* idx, int width, int height, const float* normal_map, const float* dist_map, const float* w_map,
                         float* pixels, float* normals, float* dists, float* w, void* stream);
int gs2m_mv_take_backward(int n, const long long* idx, int width, int height, const float* d_normals, const float* d_dists, float* d_normal_map,
                          float* d_dist_map, void* stream);
int gs2m_ncc_tail_forward(int n, const float* ncc, const float* w, float* out, void* workspace, void* stream);
int gs2m_ncc_tail_backward(int n, const float* ncc, const float* w, const float* out, const float* g_loss, float* d_ncc, void* stream);

/* The patch samples of multi_view_loss / roughness_loss: a random subset of exactly k set elements of a mask, every set element equally
 * likely (the reference: idx[torch.randperm(idx.numel())[:k]], utils/loss_utils.py:283-286; gs-2m_amd/gs2m_mvs.py: random_subset), in
 * element order, in two steps with counter-based random numbers (a 64-bit mix of `seed` and the element index).
 * subset_thin: element i of mask (n bytes) survives when set and u(seed, i) < (k + 4 sqrt(k)) / count(mask); the survivors' indices,
 * ascending, go to idx (capacity cap: those beyond it are dropped), counts[0] = count(mask), counts[1] = number of survivors (device ints:
 * the caller reads them); block_counts: ceil(n / 1024) ints of scratch.
 * subset_remove (m survivors > k): out[j] = idx[j + #{i < m - k : removed_i - i <= j}], removed_i = min(m - 1, floor((i + u(seed, i)) m / (m - k))):
 * one survivor removed per stratum, k kept. */
int gs2m_subset_thin(int n, const unsigned char* mask, int k, unsigned long long seed, long long* idx, int cap, int* counts, int* block_counts,
                     void* stream);
int gs2m_subset_remove(int m, int k, unsigned long long seed, const long long* idx, long long* out, void* stream);

/* out[0] = a + b * mean(x) over n contiguous floats (x 16-byte aligned): `ssim_map.mean()` (a = 0, b = 1,
 * fused_ssim/__init__.py:40-41) and the D-SSIM term lambda * (1 - ssim) of train.py:103 (a = lambda, b = -lambda). */
int gs2m_affine_mean(long long n, const float* x, float a, float b, float* out, void* workspace, void* stream);

/* plane_loss, utils/loss_utils.py:72-79: out[0] = mean over the visible Gaussians of the smallest of their three
 * scales, 0 when none is visible; out[1] = the number of visible Gaussians (kept for the backward).  scaling (P, 3): the
 * activated scales (raw = 0, `get_scaling`) or the log-scales the model stores (raw = 1, `_scaling`: the exp of
 * scene/gaussian_model.py:113-114 and its derivative are applied here).  visible: P bytes (torch.bool).  out[0] is
 * multiplied by `weight` (lambda_plane). */
int gs2m_plane_loss_forward(int P, const float* scaling, int raw, const unsigned char* visible, float weight, float* out, void* workspace,
                            void* stream);
int gs2m_plane_loss_backward(int P, const float* scaling, int raw, const unsigned char* visible, float weight, const float* out,
                             const float* g_loss, float* d_scaling, void* stream);

/* add_densification_stats, scene/gaussian_model.py:569-573, in place: for visible Gaussians grad_accum += |grad[:, :2]|,
 * grad_accum_abs += |grad[:, 2:]|, denom += 1 (viewspace_grad: (P, 4)).  With max_radii non-NULL also train.py:223-225:
 * max_radii = max(max_radii, radii) where visible and observe > 0 (observe, radii: int32). */
int gs2m_densification_stats(int P, const float* viewspace_grad, const unsigned char* visible, const int* observe, const int* radii,
                             float* grad_accum, float* grad_accum_abs, float* denom, float* max_radii, void* stream);

#ifdef __cplusplus
}
#endif
#endif
