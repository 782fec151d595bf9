// Micro-benchmark for profiles/r06_bwd_groups.md: what a 32-survivor group of the blend backward would pay on the matrix pipe and in its scans.
//   (1) v_mfma_f32_16x16x4_f32 against v_mfma_f32_32x32x2_f32 (the only fp32 shape whose A operand takes one value per lane of a
//       (32 survivors x 2 pixel columns) wave), cycles per instruction per SIMD;
//   (2) a Kogge-Stone affine scan over a DPP row of 16 (4 levels) against one over 32 lanes (4 levels + a row_bcast:15 level).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/mfma_shapes.hip -o tools/micro/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2048;

template <int KIND>  // 0: 4 x 16x16x4 per iteration, 1: 4 x 32x32x2 per iteration (two accumulators each)
__global__ void k_mfma(float* out, float a, float b) {
    v4f acc4[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
    v16f acc16[2];
    for (int i = 0; i < 16; i++) { acc16[0][i] = 0.f; acc16[1][i] = 0.f; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (KIND == 0) acc4[i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i & 1], 0, 0, 0);
            else acc16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[i & 1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) s += acc4[0][i] + acc4[1][i];
    for (int i = 0; i < 16; i++) s += acc16[0][i] + acc16[1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int LANES>  // 16: four row_shr levels; 32: + one row_bcast:15 level over the odd rows
__global__ void k_scan(float* out, float a, float b) {
    float Ax = a + threadIdx.x * 1e-6f, Ay = a, Bx = b, By = b + threadIdx.x * 1e-6f;
    for (int it = 0; it < ITERS; it++) {
#define LEVEL(D)                                                                  \
        "v_fmac_f32_dpp %2, %2, %0 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t" \
        "v_fmac_f32_dpp %3, %3, %1 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t" \
        "v_mul_f32_dpp %0, %0, %0 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t"  \
        "v_mul_f32_dpp %1, %1, %1 row_shr:" #D " row_mask:0xf bank_mask:0xf\n\t"
        if (LANES == 16) {
            asm volatile("s_nop 1\n\t" LEVEL(1) LEVEL(2) LEVEL(4) LEVEL(8) : "+v"(Ax), "+v"(Ay), "+v"(Bx), "+v"(By));
        } else {
            asm volatile("s_nop 1\n\t" LEVEL(1) LEVEL(2) LEVEL(4) LEVEL(8)
                         "v_fmac_f32_dpp %2, %2, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %3, %3, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         : "+v"(Ax), "+v"(Ay), "+v"(Bx), "+v"(By));
        }
#undef LEVEL
        Ax = Ax * 0.5f + 0.25f; Ay = Ay * 0.5f + 0.25f; Bx *= 0.5f; By *= 0.5f;  // keep the values bounded (4 plain instructions in both variants)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Ax + Ay + Bx + By;
}

// (3) issue cost of the instruction kinds of the backward's block loop at its occupancy: 16 instructions per iteration on 4 registers
template <int KIND>  // 0: v_mul_f32 (plain), 1: v_mul_f32_dpp row_shr:1, 2: v_fmac_f32_dpp row_shr:1, 3: v_pk_mul_f32, 4: v_exp_f32, 5: v_rcp_f32
__global__ void k_op(float* out, float a, float b) {
    float x0 = a + threadIdx.x * 1e-6f, x1 = a, x2 = b, x3 = b + threadIdx.x * 1e-6f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {x0, x1}, p1 = {x2, x3};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
            if (KIND == 1) asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                                        "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (KIND == 2) asm volatile("v_fmac_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                                        "v_fmac_f32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_pk_mul_f32 %1, %1, %2\n\tv_pk_mul_f32 %0, %0, %2\n\tv_pk_mul_f32 %1, %1, %2" : "+v"(p0), "+v"(p1) : "v"(p1));
            if (KIND == 4) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (KIND == 5) asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y;
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    for (int i = 0; i < 40; i++) f();  // (clock ramp)
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 4 * 4 * 256 * 4);
    const int grid = 256 * 4 * 4, block = 256;  // 4 waves per SIMD resident (the backward's occupancy), 4 rounds
    const double per = 2.4e9 * 1024 / ((double)grid * block / 64 * ITERS) * 1e-3;  // ms -> cycles per wave-iteration per SIMD
    printf("cycles per wave-iteration per SIMD (@2.4 GHz nominal, 4 waves per SIMD)\n");
    const float m16 = per * timeit([&] { k_mfma<0><<<grid, block>>>(out, 1.0001f, 0.5f); });
    const float m32 = per * timeit([&] { k_mfma<1><<<grid, block>>>(out, 1.0001f, 0.5f); });
    printf("4 x v_mfma_f32_16x16x4_f32 : %.1f  (%.1f per instruction, 64 (survivor, pixel) pairs x 16 columns each)\n", m16, m16 / 4);
    printf("4 x v_mfma_f32_32x32x2_f32 : %.1f  (%.1f per instruction, 64 pairs x 32 columns each)\n", m32, m32 / 4);
    const float s16 = per * timeit([&] { k_scan<16><<<grid, block>>>(out, 0.9f, 0.1f); });
    const float s32 = per * timeit([&] { k_scan<32><<<grid, block>>>(out, 0.9f, 0.1f); });
    printf("affine scan of 2 chains over 16 lanes (16 DPP + 4 plain) : %.1f\n", s16);
    printf("affine scan of 2 chains over 32 lanes (20 DPP + 4 plain) : %.1f\n", s32);
    const char* names[6] = {"v_mul_f32", "v_mul_f32_dpp row_shr:1", "v_fmac_f32_dpp row_shr:1", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32"};
    float t[6];
    t[0] = per * timeit([&] { k_op<0><<<grid, block>>>(out, 0.999f, 1.0001f); });
    t[1] = per * timeit([&] { k_op<1><<<grid, block>>>(out, 0.999f, 1.0001f); });
    t[2] = per * timeit([&] { k_op<2><<<grid, block>>>(out, 0.999f, 1.0001f); });
    t[3] = per * timeit([&] { k_op<3><<<grid, block>>>(out, 0.999f, 1.0001f); });
    t[4] = per * timeit([&] { k_op<4><<<grid, block>>>(out, 0.999f, 1.0001f); });
    t[5] = per * timeit([&] { k_op<5><<<grid, block>>>(out, 0.999f, 1.0001f); });
    for (int i = 0; i < 6; i++) printf("16 x %-26s: %.1f  (%.2f per instruction)\n", names[i], t[i], t[i] / 16);
    return 0;
}
