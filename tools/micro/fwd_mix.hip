// Micro-benchmark: the instruction mix of blend_fwd_q.hip's per-entry evaluation without any global memory, to
// separate (a) the vector-ALU cost of the mix itself from (b) the cost of fetching the entry through LDS broadcasts.
//   SRC 0: the entry's 20 floats sit in VGPRs (made opaque per iteration so nothing is hoisted)
//   SRC 1: read from LDS with wave-uniform addresses, as the kernel does (5 ds_read_b128 + 1 ds_read2_b64 per entry)
//   KNOCK bit 0: no colour accumulation (6 v_pk_fma_f32)   bit 1: no exp   bit 2: no tests / masks / selects
// Output: cycles per entry per SIMD at 2.4 GHz for 1 .. 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/micro/fwd_mix.hip -o tools/micro/fwd_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned long long mask_t;
constexpr int ITERS = 2048;  // chunks of 16 entries

__device__ __forceinline__ float power(float dx, float dy, float A, float B, float C) {
    const float t1 = (A * dx) * dx, t2 = (C * dy) * dy, t3 = (B * dx) * dy;
    return (-0.5f * (t1 + t2)) - t3;
}

template <int SRC, int KNOCK>
__global__ void __launch_bounds__(64) k(float* out, const float4* in) {
    __shared__ float4 s_buf[6 * 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 6 * 16; i += 64) s_buf[i] = in[i];
    __syncthreads();
    uint32_t zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const float4* const s_base = s_buf + zero;
    const float pxf = (float)(lane & 7), pyf = (float)(lane >> 3);
    float T = 1.0f;
    uint32_t last = 0;
    v2f acc[6];
    for (int q = 0; q < 6; q++) acc[q] = v2f{0.f, 0.f};
    mask_t live = ~0ull;
    int ilast = 0;
    float4 ra = in[0], rb = in[16], rc0 = in[32], rc1 = in[48], rc2 = in[64], re = in[80];
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            float4 a, b, c0, c1, c2, e;
            if (SRC == 1) {
                asm volatile("" ::: "memory");  // the kernel's buffers are rewritten per chunk: no hoisting of the reads
                a = s_base[jj]; b = s_base[16 + jj]; c0 = s_base[32 + jj]; c1 = s_base[48 + jj]; c2 = s_base[64 + jj]; e = s_base[80 + jj];
            } else {
                asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w));
                asm volatile("" : "+v"(rb.x), "+v"(rb.y));
                a = ra; b = rb; c0 = rc0; c1 = rc1; c2 = rc2; e = re;
            }
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float p2 = power(dx, dy, a.z, a.w, b.x);
            const float G = (KNOCK & 2) ? p2 : __builtin_amdgcn_exp2f(p2 * 1.4426950408889634f);
            const float alpha = fminf(0.99f, b.y * G);
            const float test_T = T * (1.0f - alpha);
            float w;
            if (!(KNOCK & 4)) {
                const mask_t cand = live & __builtin_amdgcn_ballot_w64(p2 <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
                const mask_t fin = cand & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                const mask_t contrib = cand & ~fin;
                live &= ~fin;
                if (contrib != 0ull) ilast = it * 16 + jj + 1;
                float tw;
                asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(tw) : "v"(T), "s"(contrib));
                w = alpha * tw;
                asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(last) : "v"(__float_as_uint(e.y)), "s"(contrib));
                asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(T) : "v"(test_T), "s"(contrib));
                if (live == 0ull) { T = 1.0f; live = ~0ull; }  // never taken with the data below; keeps the exit test
            } else {
                w = alpha * T;
                T = test_T * 0.5f + 0.5f;
            }
            if (!(KNOCK & 1)) {
                const v2f ww = {w, w};
                acc[0] = __builtin_elementwise_fma(v2f{c0.x, c0.y}, ww, acc[0]);
                acc[1] = __builtin_elementwise_fma(v2f{c0.z, c0.w}, ww, acc[1]);
                acc[2] = __builtin_elementwise_fma(v2f{c1.x, c1.y}, ww, acc[2]);
                acc[3] = __builtin_elementwise_fma(v2f{c1.z, c1.w}, ww, acc[3]);
                acc[4] = __builtin_elementwise_fma(v2f{c2.x, c2.y}, ww, acc[4]);
                acc[5] = __builtin_elementwise_fma(v2f{c2.z, c2.w}, ww, acc[5]);
            } else {
                acc[0].x += w;
            }
        }
    }
    float s = T + (float)last + (float)ilast;
    for (int q = 0; q < 6; q++) s += acc[q].x + acc[q].y;
    out[blockIdx.x * 64 + lane] = s;
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

template <int SRC, int KNOCK>
void run(const char* name, float* out, const float4* in) {
    printf("%-44s:", name);
    for (int wps : {1, 2, 4, 8}) {
        const int grid = 256 * 4 * wps;  // one 64-thread workgroup per wave slot
        const float ms = timeit([&] { k<SRC, KNOCK><<<grid, 64>>>(out, in); });
        printf("  %dw %.1f", wps, ms * 1e-3 * 2.4e9 * 1024 / ((double)grid * ITERS * 16));
    }
    printf("\n");
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    float4 h[6 * 16];
    for (int j = 0; j < 16; j++) {
        h[j] = make_float4(3.5f + 0.1f * j, 3.5f - 0.1f * j, 0.08f, 0.01f);     // x, y, A, B
        h[16 + j] = make_float4(0.07f, 0.02f + 0.001f * j, 0.f, 0.f);            // C, opacity (small: T never runs out)
        for (int q = 0; q < 3; q++) h[32 + 16 * q + j] = make_float4(0.1f * q, 0.2f, 0.3f + 0.01f * j, 0.4f);
        h[80 + j] = make_float4(0.f, 1e-30f * j, 0.f, 0.f);
    }
    float4* in; (void)hipMalloc(&in, sizeof(h)); (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("cycles per entry per SIMD (@2.4 GHz nominal), by waves per SIMD\n");
    run<0, 0>("registers, full mix", out, in);
    run<1, 0>("LDS broadcast reads, full mix", out, in);
    run<0, 1>("registers, no colour FMAs", out, in);
    run<1, 1>("LDS, no colour FMAs", out, in);
    run<0, 2>("registers, no exp", out, in);
    run<0, 4>("registers, no tests / selects", out, in);
    run<1, 4>("LDS, no tests / selects", out, in);
    run<0, 7>("registers, power + alpha only", out, in);
    return 0;
}
