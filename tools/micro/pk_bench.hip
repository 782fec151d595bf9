// Micro-benchmark: is packed fp32 (v_pk_fma_f32 / v_pk_mul_f32) issued at the same rate as scalar fp32 VALU on gfx950?
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/micro/pk_bench.hip -o tools/micro/pk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

__global__ void scalar_k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void packed_k(float* out, float a, float b) {
    v2f A = {a, a}, B = {b, b};
    float t = threadIdx.x;
    v2f x0 = {t, t + 1}, x1 = {t + 2, t + 3}, x2 = {t + 4, t + 5}, x3 = {t + 6, t + 7};
    v2f x4 = {t + 8, t + 9}, x5 = {t + 10, t + 11}, x6 = {t + 12, t + 13}, x7 = {t + 14, t + 15};
    for (int i = 0; i < ITERS; i++) {
        x0 = __builtin_elementwise_fma(x0, A, B); x1 = __builtin_elementwise_fma(x1, A, B);
        x2 = __builtin_elementwise_fma(x2, A, B); x3 = __builtin_elementwise_fma(x3, A, B);
        x4 = __builtin_elementwise_fma(x4, A, B); x5 = __builtin_elementwise_fma(x5, A, B);
        x6 = __builtin_elementwise_fma(x6, A, B); x7 = __builtin_elementwise_fma(x7, A, B);
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void dpp_k(float* out, float a) {  // dependent DPP chain interleaved 4-way
    float x0 = threadIdx.x * a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < ITERS; i++) {
        x0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x0), 0x111, 0xF, 0xF, false));
        x1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x1), 0x111, 0xF, 0xF, false));
        x2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x2), 0x111, 0xF, 0xF, false));
        x3 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x3), 0x111, 0xF, 0xF, false));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ void trans_k(float* out, float a) {
    float x0 = threadIdx.x * a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < ITERS; i++) {
        x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 16 * 256 * 4);
    const int grid = 256 * 8 * 4, block = 256;  // 8 waves/SIMD over the whole chip, 4 rounds
    const double winst = (double)grid * block / 64;
    float ms = timeit([&] { scalar_k<<<grid, block>>>(out, 1.0001f, 0.5f); });
    printf("scalar fma : %.3f ms  -> %.2f cycles/wave-instr/SIMD @2.4GHz\n", ms, ms * 1e-3 * 2.4e9 * 1024 / (winst * ITERS * 8));
    ms = timeit([&] { packed_k<<<grid, block>>>(out, 1.0001f, 0.5f); });
    printf("packed fma : %.3f ms  -> %.2f cycles/wave-instr/SIMD\n", ms, ms * 1e-3 * 2.4e9 * 1024 / (winst * ITERS * 8));
    ms = timeit([&] { dpp_k<<<grid, block>>>(out, 1.0001f); });
    printf("dpp add    : %.3f ms  -> %.2f cycles/wave-instr/SIMD\n", ms, ms * 1e-3 * 2.4e9 * 1024 / (winst * ITERS * 4));
    ms = timeit([&] { trans_k<<<grid, block>>>(out, 1.0001f); });
    printf("exp2/rcp   : %.3f ms  -> %.2f cycles/wave-instr/SIMD\n", ms, ms * 1e-3 * 2.4e9 * 1024 / (winst * ITERS * 4));
    return 0;
}
