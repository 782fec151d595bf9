// Throughput of global integer atomics by memory scope on a multi-XCD part, and a correctness probe of XCD-local counters:
// agent scope (coherent across the 8 L2s: executed at the memory side) against workgroup scope into a per-XCD copy selected
// by the hardware XCC_ID (executed in the XCD's own L2).  Build: hipcc -O3 --offload-arch=gfx950 atomic_scope.hip -o atomic_scope
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, width 4)
    return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
}
__device__ __forceinline__ uint32_t hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <int MODE>  // 0: agent scope, one copy; 1: workgroup scope, copy[xcc]; 2: agent scope returning; 3: workgroup scope returning, copy[xcc]
__global__ void k(uint32_t* cnt, int tiles, int per_thread, uint32_t* sink, uint32_t* xcc_hist) {
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[x & 15], 1u);
    uint32_t* base = (MODE & 1) ? cnt + (size_t)x * tiles : cnt;
    uint32_t acc = 0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_thread; k++) {
        const uint32_t t = hash(i * 131u + k) % (uint32_t)tiles;
        if (MODE == 0) __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) acc += __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) acc += __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (MODE >= 2 && acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main() {
    const int tiles = 8160, blocks = 3907, per_thread = 3;
    uint32_t *cnt, *sink, *xh;
    hipMalloc(&cnt, 16 * tiles * 4); hipMalloc(&sink, 64); hipMalloc(&xh, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f;
        unsigned long long total = 0;
        std::vector<uint32_t> h(16 * tiles), hx(16);
        for (int rep = 0; rep < 5; rep++) {
            hipMemset(cnt, 0, 16 * tiles * 4); hipMemset(xh, 0, 64);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(cnt, tiles, per_thread, sink, xh);
            if (mode == 1) k<1><<<blocks, 256>>>(cnt, tiles, per_thread, sink, xh);
            if (mode == 2) k<2><<<blocks, 256>>>(cnt, tiles, per_thread, sink, xh);
            if (mode == 3) k<3><<<blocks, 256>>>(cnt, tiles, per_thread, sink, xh);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        hipMemcpy(h.data(), cnt, 16 * tiles * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), xh, 64, hipMemcpyDeviceToHost);
        for (auto v : h) total += v;
        printf("mode %d: %.4f ms for %d atomics (%.1f G/s), counted %llu (%s); workgroups per XCC:", mode, best, blocks * 256 * per_thread,
               blocks * 256.0 * per_thread / best / 1e6, total, total == (unsigned long long)blocks * 256 * per_thread ? "ok" : "LOST UPDATES");
        for (int x = 0; x < 16; x++) printf(" %u", hx[x]);
        printf("\n");
    }
    return 0;
}
