// distCUDA2 for gfx950: mean squared distance to the 3 nearest other points.
//
// Semantics: SimpleKNN::knn, /root/reference/submodules/simple-knn/simple_knn.cu:169-204
// (Morton order :44-66, 1024-point boxes :73-108, pruned search :110-167).  The pruning
// never discards a candidate that could enter the 3-best set, so the output is the exact
// 3-NN mean with the reference's arithmetic (dx*dx + dy*dy + dz*dz, ascending best[3],
// (b0+b1+b2)/3.0f); compiled with -ffp-contract=off.
// MI355X-first differences: no host round trips (the AABB stays on the device; the
// reference copies min/max back twice), points are gathered once into Morton order as
// float4 so every later read is a coalesced stream, and a 256-thread workgroup decides per
// box with one ballot whether ANY of its points needs it, then stages the box's 1024
// points in LDS once for the whole workgroup instead of 1024 dependent gathers per thread.
#include "common.h"
#include <cfloat>

namespace {

constexpr int BOX = 1024;

struct MinMax {
    float mnx, mny, mnz, mxx, mxy, mxz;
};

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// stage 1: per-block AABB; stage 2 (one block): fold partials.  The reduction is seeded with
// the origin for both min and max, as the reference's init = {0,0,0} does (simple_knn.cu:174-183).
__global__ void __launch_bounds__(256) aabb_kernel(int n, const float* __restrict__ pts, int stride, int final_pass,
                                                   MinMax* __restrict__ out) {
    __shared__ MinMax s[4];
    MinMax m = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!final_pass) {
            const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
            m.mnx = fminf(m.mnx, x); m.mny = fminf(m.mny, y); m.mnz = fminf(m.mnz, z);
            m.mxx = fmaxf(m.mxx, x); m.mxy = fmaxf(m.mxy, y); m.mxz = fmaxf(m.mxz, z);
        } else {
            const MinMax o = reinterpret_cast<const MinMax*>(pts)[i];
            m.mnx = fminf(m.mnx, o.mnx); m.mny = fminf(m.mny, o.mny); m.mnz = fminf(m.mnz, o.mnz);
            m.mxx = fmaxf(m.mxx, o.mxx); m.mxy = fmaxf(m.mxy, o.mxy); m.mxz = fmaxf(m.mxz, o.mxz);
        }
    }
    m.mnx = wave_min(m.mnx); m.mny = wave_min(m.mny); m.mnz = wave_min(m.mnz);
    m.mxx = wave_max(m.mxx); m.mxy = wave_max(m.mxy); m.mxz = wave_max(m.mxz);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    gs2m_sync();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            m.mnx = fminf(m.mnx, s[w].mnx); m.mny = fminf(m.mny, s[w].mny); m.mnz = fminf(m.mnz, s[w].mnz);
            m.mxx = fmaxf(m.mxx, s[w].mxx); m.mxy = fmaxf(m.mxy, s[w].mxy); m.mxz = fmaxf(m.mxz, s[w].mxz);
        }
        out[blockIdx.x] = m;
    }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ void morton_kernel(int P, const float* __restrict__ pts, const MinMax* __restrict__ aabb,
                              uint32_t* __restrict__ codes) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const MinMax b = *aabb;
    const float x = pts[3 * (size_t)idx], y = pts[3 * (size_t)idx + 1], z = pts[3 * (size_t)idx + 2];
    const uint32_t mx = prep_morton((uint32_t)(((x - b.mnx) / (b.mxx - b.mnx)) * ((1 << 10) - 1)));
    const uint32_t my = prep_morton((uint32_t)(((y - b.mny) / (b.mxy - b.mny)) * ((1 << 10) - 1)));
    const uint32_t mz = prep_morton((uint32_t)(((z - b.mnz) / (b.mxz - b.mnz)) * ((1 << 10) - 1)));
    codes[idx] = mx | (my << 1) | (mz << 2);
}

__global__ void gather_points_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                     float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = order[i];
    sorted[i] = make_float4(pts[3 * (size_t)g], pts[3 * (size_t)g + 1], pts[3 * (size_t)g + 2], 0.f);
}

// AABB of each run of 1024 Morton-sorted points (boxMinMax, simple_knn.cu:73-108)
__global__ void __launch_bounds__(256) box_kernel(int P, const float4* __restrict__ sorted, MinMax* __restrict__ boxes) {
    __shared__ MinMax s[4];
    MinMax m = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    const int start = blockIdx.x * BOX;
    for (int k = threadIdx.x; k < BOX; k += 256) {
        const int i = start + k;
        if (i < P) {
            const float4 p = sorted[i];
            m.mnx = fminf(m.mnx, p.x); m.mny = fminf(m.mny, p.y); m.mnz = fminf(m.mnz, p.z);
            m.mxx = fmaxf(m.mxx, p.x); m.mxy = fmaxf(m.mxy, p.y); m.mxz = fmaxf(m.mxz, p.z);
        }
    }
    m.mnx = wave_min(m.mnx); m.mny = wave_min(m.mny); m.mnz = wave_min(m.mnz);
    m.mxx = wave_max(m.mxx); m.mxy = wave_max(m.mxy); m.mxz = wave_max(m.mxz);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    gs2m_sync();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            m.mnx = fminf(m.mnx, s[w].mnx); m.mny = fminf(m.mny, s[w].mny); m.mnz = fminf(m.mnz, s[w].mnz);
            m.mxx = fmaxf(m.mxx, s[w].mxx); m.mxy = fmaxf(m.mxy, s[w].mxy); m.mxz = fmaxf(m.mxz, s[w].mxz);
        }
        boxes[blockIdx.x] = m;
    }
}

__device__ __forceinline__ void update3(const float4& ref, const float4& p, float* best) {
    const float dx = p.x - ref.x, dy = p.y - ref.y, dz = p.z - ref.z;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
    }
}

__device__ __forceinline__ float dist_box_point(const MinMax& b, const float4& p) {
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (p.x < b.mnx || p.x > b.mxx) dx = fminf(fabsf(p.x - b.mnx), fabsf(p.x - b.mxx));
    if (p.y < b.mny || p.y > b.mxy) dy = fminf(fabsf(p.y - b.mny), fabsf(p.y - b.mxy));
    if (p.z < b.mnz || p.z > b.mxz) dz = fminf(fabsf(p.z - b.mnz), fabsf(p.z - b.mxz));
    return dx * dx + dy * dy + dz * dz;
}

// boxMeanDist (simple_knn.cu:134-167)
__global__ void __launch_bounds__(256) knn_kernel(int P, const float4* __restrict__ sorted,
                                                  const uint32_t* __restrict__ order, const MinMax* __restrict__ boxes,
                                                  int num_boxes, float* __restrict__ dists) {
    __shared__ float4 s_pts[BOX];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < P;
    float4 point = make_float4(0.f, 0.f, 0.f, 0.f);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    if (valid) {
        point = sorted[idx];
        for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
            if (i == idx) continue;
            update3(point, sorted[i], best);
        }
    }
    const float reject = best[2];
    best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
    for (int b = 0; b < num_boxes; b++) {
        const MinMax box = boxes[b];
        bool want = false;
        if (valid) {
            const float d = dist_box_point(box, point);
            want = !(d > reject || d > best[2]);
        }
        if (gs2m_sync_or(want)) {
            const int start = b * BOX, n = min(BOX, P - start);
            for (int k = threadIdx.x; k < n; k += 256) s_pts[k] = sorted[start + k];
            gs2m_sync();
            if (want) {
                for (int k = 0; k < n; k++) {
                    if (start + k == idx) continue;
                    update3(point, s_pts[k], best);
                }
            }
        }
        gs2m_sync();
    }
    if (valid) dists[order[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

extern "C" int gs2m_knn_dist2(int P, const float* points, float* mean_dists, gs2m_alloc_fn scratch_alloc,
                              void* scratch_user, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (P < 0) return GS2M_ERR_INVALID_ARG;
    if (P == 0) return GS2M_OK;
    if (!points || !mean_dists || !scratch_alloc) return GS2M_ERR_INVALID_ARG;
    const size_t n = (size_t)P;
    const int num_boxes = (P + BOX - 1) / BOX;
    const int nblk = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    const size_t sort_bytes = gs2m_radix_temp_bytes(n, 32);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = gs2m_align_up(off + bytes); return o; };
    const size_t o_codes = take(n * 4), o_iota = take(n * 4), o_codes_s = take(n * 4), o_order = take(n * 4);
    const size_t o_kA = take(n * 4);
    const size_t o_sorted = take(n * sizeof(float4)), o_boxes = take((size_t)num_boxes * sizeof(MinMax));
    const size_t o_part = take((size_t)nblk * sizeof(MinMax)), o_aabb = take(sizeof(MinMax)), o_temp = take(sort_bytes);
    char* base = scratch_alloc(off + GS2M_ALIGN, scratch_user);
    if (!base) return GS2M_ERR_ALLOC;
    base = (char*)gs2m_align_up((size_t)(uintptr_t)base);
    uint32_t* codes = (uint32_t*)(base + o_codes); uint32_t* iota = (uint32_t*)(base + o_iota);
    uint32_t* codes_s = (uint32_t*)(base + o_codes_s); uint32_t* order = (uint32_t*)(base + o_order);
    float4* sorted = (float4*)(base + o_sorted); MinMax* boxes = (MinMax*)(base + o_boxes);
    MinMax* part = (MinMax*)(base + o_part); MinMax* aabb = (MinMax*)(base + o_aabb);

    aabb_kernel<<<nblk, 256, 0, s>>>(P, points, 3, 0, part);
    aabb_kernel<<<1, 256, 0, s>>>(nblk, reinterpret_cast<const float*>(part), 0, 1, aabb);
    morton_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, points, aabb, codes);
    // 30-bit Morton codes; values implicit (index); iota doubles as the ping value buffer
    if (gs2m_radix_sort_pairs(base + o_temp, sort_bytes, codes, nullptr, (uint32_t*)(base + o_kA), iota, codes_s, order, n, 32, false, s) != hipSuccess) return GS2M_ERR_HIP;
    gather_points_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, points, order, sorted);
    box_kernel<<<num_boxes, 256, 0, s>>>(P, sorted, boxes);
    knn_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, sorted, order, boxes, num_boxes, mean_dists);
    if (hipGetLastError() != hipSuccess) return GS2M_ERR_HIP;
    return GS2M_OK;
}
