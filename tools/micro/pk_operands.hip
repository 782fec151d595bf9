// Micro-benchmark: what does a packed fp32 instruction cost on gfx950 as a function of (a) how many of its sources are
// per-lane VGPR pairs, (b) whether consecutive instructions of a wave depend on each other, (c) waves per SIMD?
// Every kernel runs the same number of wave-instructions; the table is cycles per wave-instruction per SIMD at 2.4 GHz.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/micro/pk_operands.hip -o tools/micro/pk_operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

// KIND 0: plain v_fma_f32, 3 VGPR sources        1: v_pk_fma_f32, 1 VGPR pair + 2 uniform
//      2: v_pk_fma_f32, 3 VGPR pairs             3: v_pk_mul_f32, 2 VGPR pairs
//      4: v_pk_add_f32, 2 VGPR pairs             5: plain v_mul_f32, 2 VGPRs
// CHAINS independent accumulators per thread (1 = every instruction depends on the previous one)
template <int KIND, int CHAINS>
__global__ void k(float* out, float a, float b) {
    const float t = (float)threadIdx.x * 1e-3f;
    v2f x[CHAINS], y[CHAINS], z[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        x[c] = v2f{t + c, t + c + 0.5f};
        y[c] = v2f{1.0f + t * 1e-3f + c * 1e-4f, 1.0f - t * 1e-3f - c * 1e-4f};
        z[c] = v2f{t * 0.25f + c, t * 0.125f - c};
    }
    const v2f A = {a, a}, B = {b, b};
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int rep = 0; rep < 8 / CHAINS; rep++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) x[c].x = __builtin_fmaf(x[c].x, y[c].x, z[c].x);
                if (KIND == 1) x[c] = __builtin_elementwise_fma(x[c], A, B);
                if (KIND == 2) x[c] = __builtin_elementwise_fma(x[c], y[c], z[c]);
                if (KIND == 3) x[c] = x[c] * y[c];
                if (KIND == 4) x[c] = x[c] + y[c];
                if (KIND == 5) x[c].x = x[c].x * y[c].x;
            }
        }
    }
    v2f s = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
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

template <int KIND, int CHAINS>
void run(const char* name, float* out) {
    printf("%-34s chains %d :", name, CHAINS);
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: 256 CUs x 4 SIMDs x wps waves, one round
        const int block = 64 * wps * 4 > 1024 ? 1024 : 64 * wps * 4;  // one workgroup per CU up to 16 waves, two beyond
        const int grid = 256 * (64 * wps * 4) / block;
        const double winst = (double)grid * block / 64;
        const float ms = timeit([&] { k<KIND, CHAINS><<<grid, block>>>(out, 1.0001f, 0.5f); });
        printf("  %dw %.2f", wps, ms * 1e-3 * 2.4e9 * 1024 / (winst * ITERS * 8));
    }
    printf("\n");
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 2048 * 4);
    printf("cycles per wave-instruction per SIMD (@2.4 GHz nominal), by waves per SIMD\n");
    run<0, 8>("v_fma_f32 3 vgpr", out);            run<0, 1>("v_fma_f32 3 vgpr", out);          run<0, 2>("v_fma_f32 3 vgpr", out);
    run<5, 8>("v_mul_f32 2 vgpr", out);            run<5, 1>("v_mul_f32 2 vgpr", out);
    run<1, 8>("v_pk_fma_f32 1 pair + uniform", out); run<1, 1>("v_pk_fma_f32 1 pair + uniform", out);
    run<2, 8>("v_pk_fma_f32 3 pairs", out);        run<2, 1>("v_pk_fma_f32 3 pairs", out);      run<2, 2>("v_pk_fma_f32 3 pairs", out);
    run<3, 8>("v_pk_mul_f32 2 pairs", out);        run<3, 1>("v_pk_mul_f32 2 pairs", out);
    run<4, 8>("v_pk_add_f32 2 pairs", out);        run<4, 1>("v_pk_add_f32 2 pairs", out);
    return 0;
}
