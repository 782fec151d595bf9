// Micro-benchmark: does a kernel on a non-blocking side stream (forked/joined with events) overlap a chain of
// small kernels on the caller's stream (null stream or a created stream)?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/stream_overlap.hip -o tools/micro/stream_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* out, int iters) {  // few blocks, long: "latency-bound chain" stand-in
    float x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void stream_copy(const float4* a, float4* b, size_t n) {  // HBM-bound
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    float* o; float4 *a, *b;
    const size_t n = 16u << 20;  // 256 MB each
    (void)hipMalloc(&o, 1 << 20); (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16);
    hipStream_t side, mainS;
    (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    (void)hipStreamCreate(&mainS);
    hipEvent_t f, j, t0, t1;
    (void)hipEventCreateWithFlags(&f, hipEventDisableTiming); (void)hipEventCreateWithFlags(&j, hipEventDisableTiming);
    (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    for (int which = 0; which < 2; which++) {
        hipStream_t s = which ? mainS : (hipStream_t)0;
        for (int mode = 0; mode < 3; mode++) {  // 0: chain only, 1: serial chain + copy, 2: copy on the side stream
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(t0, s);
                if (mode == 2) {
                    (void)hipEventRecord(f, s); (void)hipStreamWaitEvent(side, f, 0);
                    stream_copy<<<2048, 256, 0, side>>>(a, b, n);
                }
                for (int k = 0; k < 8; k++) spin<<<64, 256, 0, s>>>(o, 20000);
                if (mode == 1) stream_copy<<<2048, 256, 0, s>>>(a, b, n);
                if (mode == 2) { (void)hipEventRecord(j, side); (void)hipStreamWaitEvent(s, j, 0); }
                spin<<<64, 256, 0, s>>>(o, 10);
                (void)hipEventRecord(t1, s); (void)hipEventSynchronize(t1);
                float ms; (void)hipEventElapsedTime(&ms, t0, t1);
                best = ms < best ? ms : best;
            }
            printf("%s stream, mode %d (%s): %.3f ms\n", which ? "created" : "null", mode,
                   mode == 0 ? "chain only" : mode == 1 ? "chain + copy serial" : "copy on side stream", best);
        }
    }
    return 0;
}
