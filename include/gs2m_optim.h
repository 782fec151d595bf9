/* gs2m_optim.h -- C ABI of the fused optimizer step, part of libgs2m_raster.so.
 *
 * Replaces, for the reference's Gaussian parameter groups, the call
 *     gaussians.optimizer.step()                                  train.py:258
 * of the optimizer built at scene/gaussian_model.py:230-245
 *     torch.optim.Adam(l, lr=0.0, eps=1e-15)      (nine groups: xyz, f_dc, f_rest, opacity, scaling, rotation,
 *                                                  albedo, roughness, metallic; per-group lr)
 * i.e. torch/optim/adam.py with weight_decay = 0, amsgrad = False, maximize = False.  One kernel launch updates up
 * to GS2M_ADAM_MAX_TENSORS tensors in place in a single pass (28 B of HBM traffic per element instead of the ~100 B
 * of the eight-pass foreach implementation), in torch's arithmetic order, so the results match torch.optim.Adam to
 * the last bit on the same device (tests/test_optim_gpu.py).
 *
 * All pointers are DEVICE pointers to contiguous fp32 arrays of `numel` elements.  No torch types; the Python
 * mirror (gs-2m_amd/gs2m_optim.py: class Adam, a torch.optim.Optimizer with torch.optim.Adam's state layout, so
 * the reference's densification code that edits optimizer.state[...]["exp_avg"/"exp_avg_sq"] keeps working) binds
 * this through ctypes.  There is no CPU path. */
#ifndef GS2M_OPTIM_H
#define GS2M_OPTIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS2M_ADAM_MAX_TENSORS 16 /* per kernel launch; more tensors = more launches, same call */

typedef struct gs2m_adam_tensor {
    float* param;        /* updated in place */
    const float* grad;   /* read only */
    float* exp_avg;      /* first moment, updated in place  (optimizer.state[p]["exp_avg"]) */
    float* exp_avg_sq;   /* second moment, updated in place (optimizer.state[p]["exp_avg_sq"]) */
    uint64_t numel;
    double lr;           /* param_group["lr"] */
    int64_t step;        /* optimizer.state[p]["step"] AFTER its increment for this update (>= 1) */
} gs2m_adam_tensor;

/* One Adam update of every tensor in `tensors` (host array, read before the call returns), asynchronous on
 * `stream`.  beta1, beta2, eps: param_group["betas"], param_group["eps"].  Returns GS2M_OK (0) or a negative
 * GS2M_ERR_* code (gs2m_raster.h). */
int gs2m_adam_step(int n_tensors, const gs2m_adam_tensor* tensors, double beta1, double beta2, double eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
