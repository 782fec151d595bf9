// Fused multi-tensor Adam for the reference's nine parameter groups (the reference builds
// torch.optim.Adam(l, lr=0.0, eps=1e-15) at scene/gaussian_model.py:245 and steps it at train.py:258).
//
// torch's default (foreach) Adam runs eight elementwise passes over every tensor -- lerp, mul, addcmul, sqrt, div,
// add, addcdiv, plus the temporaries -- 100 B of HBM traffic per element.  One pass needs 28 B: read param, grad,
// exp_avg, exp_avg_sq, write the three that change.  Everything here is that one pass, for up to GS2M_ADAM_MAX_TENSORS
// tensors per launch (one launch for the whole model), in the arithmetic order of torch/optim/adam.py
// `_multi_tensor_adam` (weight_decay = 0, amsgrad = False, maximize = False) with its scalar factors formed in
// double precision on the host exactly as the Python code forms them:
//     exp_avg     = lerp(exp_avg, grad, 1 - beta1)                         = fma(1 - beta1, grad - exp_avg, exp_avg)
//     exp_avg_sq  = fma(1 - beta2, grad * grad, exp_avg_sq * beta2)
//     denom       = sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps
//     param       = fma(-(lr / (1 - beta1^step)), exp_avg / denom, param)
// HBM bound: 28 B per element, nothing else.
#include "common.h"
#include "../../include/gs2m_optim.h"

#include <cmath>

namespace {

constexpr int ADAM_THREADS = 256;
constexpr int ADAM_VEC_PER_THREAD = 4;                                    // float4 per thread
constexpr int ADAM_CHUNK = ADAM_THREADS * ADAM_VEC_PER_THREAD * 4;         // elements per workgroup

struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    unsigned long long n;
    float neg_step_size;  // -(lr / bias_correction1)
    float bc2_sqrt;       // sqrt(bias_correction2)
    unsigned int first_block;
    unsigned int vec_ok;  // all four pointers 16-byte aligned
};

struct AdamLaunch {
    AdamTensor t[GS2M_ADAM_MAX_TENSORS];
    int count;
    float w1;     // 1 - beta1
    float beta2;
    float w2;     // 1 - beta2
    float eps;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamLaunch& L, float nss, float bc2s) {
    m = __builtin_fmaf(L.w1, g - m, m);
    v = __builtin_fmaf(L.w2, g * g, v * L.beta2);
    const float denom = sqrtf(v) / bc2s + L.eps;
    p = __builtin_fmaf(nss, m / denom, p);
}

template <bool NT>  // non-temporal access for everything but p's store (see below): the launcher's choice by the size of the step
__global__ void __launch_bounds__(ADAM_THREADS) adam_kernel(const AdamLaunch L) {
    // which tensor: a scalar walk over <= 16 block offsets
    int ti = 0;
#pragma unroll 1
    for (int k = 1; k < L.count; k++)
        if (blockIdx.x >= L.t[k].first_block) ti = k;
    const AdamTensor T = L.t[ti];
    const unsigned long long base = (unsigned long long)(blockIdx.x - T.first_block) * ADAM_CHUNK;
    const float nss = T.neg_step_size, bc2s = T.bc2_sqrt;
    if (T.vec_ok && base + ADAM_CHUNK <= T.n) {
        float4 p[ADAM_VEC_PER_THREAD], g[ADAM_VEC_PER_THREAD], m[ADAM_VEC_PER_THREAD], v[ADAM_VEC_PER_THREAD];
        const size_t q0 = base / 4 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < ADAM_VEC_PER_THREAD; u++) {
            const size_t q = q0 + (size_t)u * ADAM_THREADS;
            // g, m, v: read here and not again before the next step; p: read again by the next view's preprocess kernel (its
            // STORE below is the one plain access).  nt on all four loads + the m / v stores: 0.301 -> 0.264 ms at 1M Gaussians
            // (6.8 TB/s of the 28 B per element), the training iteration 1.866 -> 1.810 ms; with p's store nt as well: 0.287.
            // At C4's size (420 k Gaussians: a tensor set is 99 MB, the step's 0.7 GB partly live in the 256-MB Infinity Cache from
            // one iteration to the next) the hint LOSES 7 %: the launcher takes it from 160 MB per tensor set on.
            if (NT) {
                g[u] = gs2m_ldnt(reinterpret_cast<const float4*>(T.g) + q);
                p[u] = gs2m_ldnt(reinterpret_cast<const float4*>(T.p) + q);
                m[u] = gs2m_ldnt(reinterpret_cast<const float4*>(T.m) + q);
                v[u] = gs2m_ldnt(reinterpret_cast<const float4*>(T.v) + q);
            } else {
                g[u] = reinterpret_cast<const float4*>(T.g)[q];
                p[u] = reinterpret_cast<const float4*>(T.p)[q];
                m[u] = reinterpret_cast<const float4*>(T.m)[q];
                v[u] = reinterpret_cast<const float4*>(T.v)[q];
            }
        }
        // An element whose gradient and both moments are zero comes out bit-identical (p + nss * 0 / eps = p): the
        // Gaussians a run has never seen, and every SH band above the active degree until the schedule reaches it
        // (train.py:81-82: 45 of a Gaussian's 62 parameters for the first 1000 iterations).  Their 12 of 28 bytes of
        // stores are skipped -- exact, whatever the history: the test is on the bits of the quad before and after.
        bool same[ADAM_VEC_PER_THREAD];
        auto bits_equal = [](const float4 a, const float4 b) {
            return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
                   __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
        };
#pragma unroll
        for (int u = 0; u < ADAM_VEC_PER_THREAD; u++) {
            const float4 p0 = p[u], m0 = m[u], v0 = v[u];
            adam_one(p[u].x, g[u].x, m[u].x, v[u].x, L, nss, bc2s);
            adam_one(p[u].y, g[u].y, m[u].y, v[u].y, L, nss, bc2s);
            adam_one(p[u].z, g[u].z, m[u].z, v[u].z, L, nss, bc2s);
            adam_one(p[u].w, g[u].w, m[u].w, v[u].w, L, nss, bc2s);
            same[u] = bits_equal(p0, p[u]) && bits_equal(m0, m[u]) && bits_equal(v0, v[u]);
        }
#pragma unroll
        for (int u = 0; u < ADAM_VEC_PER_THREAD; u++) {
            const size_t q = q0 + (size_t)u * ADAM_THREADS;
            if (!same[u]) {
                reinterpret_cast<float4*>(T.p)[q] = p[u];
                if (NT) {
                    gs2m_stnt(reinterpret_cast<float4*>(T.m) + q, m[u]);
                    gs2m_stnt(reinterpret_cast<float4*>(T.v) + q, v[u]);
                } else {
                    reinterpret_cast<float4*>(T.m)[q] = m[u];
                    reinterpret_cast<float4*>(T.v)[q] = v[u];
                }
            }
        }
    } else {  // ragged tail or an unaligned tensor
        for (unsigned long long e = base + threadIdx.x; e < base + ADAM_CHUNK && e < T.n; e += ADAM_THREADS) {
            float p = T.p[e], m = T.m[e], v = T.v[e];
            adam_one(p, T.g[e], m, v, L, nss, bc2s);
            T.p[e] = p; T.m[e] = m; T.v[e] = v;
        }
    }
}

}  // namespace

extern "C" int gs2m_adam_step(int n_tensors, const gs2m_adam_tensor* tensors, double beta1, double beta2, double eps,
                              void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return GS2M_ERR_INVALID_ARG;
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return GS2M_ERR_INVALID_ARG;
    for (int i = 0; i < n_tensors; i++) {
        const gs2m_adam_tensor& t = tensors[i];
        if (t.step < 1 || (t.numel > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq))) return GS2M_ERR_INVALID_ARG;
    }
    for (int i0 = 0; i0 < n_tensors; i0 += GS2M_ADAM_MAX_TENSORS) {
        AdamLaunch L;
        L.count = 0;
        // the scalar factors, in double as torch/optim/adam.py forms them, then rounded once to fp32
        L.w1 = (float)(1.0 - beta1);
        L.beta2 = (float)beta2;
        L.w2 = (float)(1.0 - beta2);
        L.eps = (float)eps;
        unsigned long long blocks = 0;
        for (int i = i0; i < n_tensors && i < i0 + GS2M_ADAM_MAX_TENSORS; i++) {
            const gs2m_adam_tensor& t = tensors[i];
            if (t.numel == 0) continue;
            AdamTensor& T = L.t[L.count++];
            T.p = t.param; T.g = t.grad; T.m = t.exp_avg; T.v = t.exp_avg_sq; T.n = t.numel;
            const double bc1 = 1.0 - std::pow(beta1, (double)t.step);
            const double bc2 = 1.0 - std::pow(beta2, (double)t.step);
            T.neg_step_size = (float)((t.lr / bc1) * -1.0);
            T.bc2_sqrt = (float)std::sqrt(bc2);
            T.first_block = (unsigned int)blocks;
            T.vec_ok = (((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0;
            blocks += (t.numel + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (blocks > 0x7fffffffull) return GS2M_ERR_UNSUPPORTED;
        }
        if (L.count == 0) continue;
        unsigned long long elements = 0;
        for (int k = 0; k < L.count; k++) elements += L.t[k].n;
        if (elements * 4ull > 160ull * 1024 * 1024)
            adam_kernel<true><<<(unsigned int)blocks, ADAM_THREADS, 0, (hipStream_t)stream>>>(L);
        else
            adam_kernel<false><<<(unsigned int)blocks, ADAM_THREADS, 0, (hipStream_t)stream>>>(L);
        if (hipGetLastError() != hipSuccess) return GS2M_ERR_HIP;
    }
    return GS2M_OK;
}
