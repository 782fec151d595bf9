// Micro-benchmark: do fp32 MFMA and ordinary VALU instructions overlap on one SIMD of gfx950?
// Three kernels with the same loop count: VALU only, MFMA only, both interleaved (independent data).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/mfma_overlap.hip -o tools/micro/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;

template <int NV, int NM, int KIND>  // per iteration: NV fmas, NM mfmas; KIND 0: 16x16x4, 1: 4x4x1
__global__ void k(float* out, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    v4f acc[4];
    for (int i = 0; i < 4; i++) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NV; i++) x[i & 7] = __builtin_fmaf(x[i & 7], a, b);
#pragma unroll
        for (int i = 0; i < NM; i++) {
            if (KIND == 0) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
            else acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += x[i];
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 4 * 256 * 4);
    const int grid = 256 * 8 * 4, block = 256;  // 8 waves/SIMD, 4 rounds
    const double per = 2.4e9 * 1024 / ((double)grid * block / 64 * ITERS) * 1e-3;  // ms -> cycles per wave-iteration per SIMD
    printf("cycles per wave-iteration per SIMD (@2.4 GHz nominal)\n");
    printf("32 fma                  : %.1f\n", per * timeit([&] { k<32, 0, 0><<<grid, block>>>(out, 1.0001f, 0.5f); }));
    printf("4 mfma16x16x4           : %.1f\n", per * timeit([&] { k<0, 4, 0><<<grid, block>>>(out, 1.0001f, 0.5f); }));
    printf("32 fma + 4 mfma16x16x4  : %.1f\n", per * timeit([&] { k<32, 4, 0><<<grid, block>>>(out, 1.0001f, 0.5f); }));
    printf("4 mfma4x4x1             : %.1f\n", per * timeit([&] { k<0, 4, 1><<<grid, block>>>(out, 1.0001f, 0.5f); }));
    printf("32 fma + 4 mfma4x4x1    : %.1f\n", per * timeit([&] { k<32, 4, 1><<<grid, block>>>(out, 1.0001f, 0.5f); }));
    printf("8 fma + 4 mfma16x16x4   : %.1f\n", per * timeit([&] { k<8, 4, 0><<<grid, block>>>(out, 1.0001f, 0.5f); }));
    return 0;
}
