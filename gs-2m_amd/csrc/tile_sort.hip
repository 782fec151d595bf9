// Per-tile (depth, Gaussian id) sort on chip + the second binning level, for gfx950.
//
// binning.hip buckets the instances by tile (a counting sort): ranges[tile] is final before anything is placed, but the
// order inside a tile's span is the arrival order of fill_kernel's atomics.  This file gives every span the reference's
// order -- ascending view depth, ties in Gaussian-id order: what its 45-bit radix sort of (tile << 32 | depth) keys over the
// id-ordered duplicateWithKeys output produces (rasterizer_impl.cu:288-296) -- and then splits the sorted
// list into the four per-quadrant lists the blend kernels walk (what quad_lists_kernel did in rounds 2-4).
//   * spans of up to 1024 entries (every tile of the bench scenes): ONE WAVE per tile, the span in registers as (depth, local
//     index) pairs, lane-major (lane l holds elements l E .. l E + E - 1, E = 8 or 16), sorted by a bitonic network in its
//     all-ascending form (each merge starts with a mirrored compare, so that padding with +inf needs no direction bits):
//     strides inside a lane are register renaming + compare-exchange, strides across lanes are DPP moves (quad_perm,
//     row_half_mirror / row_mirror, row_ror:8, a masked row_shl:4 / row_shr:4 pair) and, for the three strides that cross a
//     DPP row, ds_bpermute; levels above the span's length are skipped (their input is already in order);
//   * equal depths (two Gaussians at exactly the same view depth) are put in id order afterwards -- a rare wave-uniform
//     slow path: every element of a run of equal keys counts the members with a smaller id;
//   * the sorted ids / first rows go through LDS once (lane-major -> position-major), then ballots give every instance its
//     place in each of the four quadrant lists with coalesced stores; the gradient row of a list entry = the instance's first
//     row (relative to its emit wave, fill_kernel) + the wave's base (rowscan_kernel) + its quadrants before this one;
//   * longer spans: a workgroup per tile, the same network over LDS (up to 4096 entries) or, beyond, over the tile's own
//     (still unused) quadrant-list region in global memory -- slow, correct, exercised by the dense-scene tests.
#include "common.h"

namespace {

// ---- cross-lane moves ---------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
// value of lane ^ (1 << B)
template <int B>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
    if constexpr (B == 0) return dpp_u32<DPP_QUAD_PERM(1, 0, 3, 2)>(v, v);
    else if constexpr (B == 1) return dpp_u32<DPP_QUAD_PERM(2, 3, 0, 1)>(v, v);
    else if constexpr (B == 2) {
        // banks (groups of 4 lanes) 0 and 2 of every row read 4 lanes up, banks 1 and 3 read 4 lanes down
        const uint32_t t = dpp_u32<0x104 /* row_shl:4 */, 0xF, 0x5>(v, v);
        return dpp_u32<0x114 /* row_shr:4 */, 0xF, 0xA>(t, v);
    } else if constexpr (B == 3) return dpp_u32<0x128 /* row_ror:8 */>(v, v);
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ (1 << B)) << 2, (int)v);
}
// value of lane ^ ((1 << T) - 1): mirrored inside groups of 2^T lanes
template <int T>
__device__ __forceinline__ uint32_t lane_flip(uint32_t v, int lane) {
    if constexpr (T == 1) return dpp_u32<DPP_QUAD_PERM(1, 0, 3, 2)>(v, v);
    else if constexpr (T == 2) return dpp_u32<DPP_QUAD_PERM(3, 2, 1, 0)>(v, v);
    else if constexpr (T == 3) return dpp_u32<DPP_ROW_HALF_MIRROR>(v, v);
    else if constexpr (T == 4) return dpp_u32<DPP_ROW_MIRROR>(v, v);
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ ((1 << T) - 1)) << 2, (int)v);
}

// ---- the network, element i = lane * E + e --------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void ce(uint32_t (&k)[E], uint32_t (&x)[E], int a, int b) {  // a < b: the smaller key to a
    const bool sw = k[b] < k[a];
    const uint32_t ka = sw ? k[b] : k[a], kb = sw ? k[a] : k[b], xa = sw ? x[b] : x[a], xb = sw ? x[a] : x[b];
    k[a] = ka; k[b] = kb; x[a] = xa; x[b] = xb;
}
template <int E, int KB>  // mirrored compare inside blocks of 2^KB elements of one lane
__device__ __forceinline__ void inlane_flip(uint32_t (&k)[E], uint32_t (&x)[E]) {
#pragma unroll
    for (int e = 0; e < E; e++)
        if (((e >> (KB - 1)) & 1) == 0) ce<E>(k, x, e, e ^ ((1 << KB) - 1));
}
template <int E, int JB>
__device__ __forceinline__ void inlane_xor(uint32_t (&k)[E], uint32_t (&x)[E]) {
#pragma unroll
    for (int e = 0; e < E; e++)
        if (((e >> JB) & 1) == 0) ce<E>(k, x, e, e | (1 << JB));
}
// Across lanes: the lane with the lower number keeps the smaller keys.  On equal keys both keep their own (consistent on
// both sides; the order of equal depths is settled afterwards).
template <int E, int T>
__device__ __forceinline__ void cross_flip(uint32_t (&k)[E], uint32_t (&x)[E], int lane) {
    const bool lower = ((lane >> (T - 1)) & 1) == 0;
    uint32_t nk[E], nx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t ok = lane_flip<T>(k[E - 1 - e], lane), ox = lane_flip<T>(x[E - 1 - e], lane);
        const bool take = lower ? (ok < k[e]) : (ok > k[e]);
        nk[e] = take ? ok : k[e];
        nx[e] = take ? ox : x[e];
    }
#pragma unroll
    for (int e = 0; e < E; e++) { k[e] = nk[e]; x[e] = nx[e]; }
}
template <int E, int B>
__device__ __forceinline__ void cross_xor(uint32_t (&k)[E], uint32_t (&x)[E], int lane) {
    const bool lower = ((lane >> B) & 1) == 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t ok = lane_xor<B>(k[e], lane), ox = lane_xor<B>(x[e], lane);
        const bool take = lower ? (ok < k[e]) : (ok > k[e]);
        k[e] = take ? ok : k[e];
        x[e] = take ? ox : x[e];
    }
}
template <int E, int LE, int JB>  // compare-exchange steps with strides 2^JB ... 2^0
__device__ __forceinline__ void xor_steps(uint32_t (&k)[E], uint32_t (&x)[E], int lane) {
    if constexpr (JB >= 0) {
        if constexpr (JB >= LE) cross_xor<E, JB - LE>(k, x, lane);
        else inlane_xor<E, JB>(k, x);
        xor_steps<E, LE, JB - 1>(k, x, lane);
    }
}
template <int E, int LE, int KB>  // merge levels KB ... LE + 6 (sorted blocks of 2^(KB-1) -> 2^KB); a level whose blocks are longer than the data is a no-op
__device__ __forceinline__ void levels(uint32_t (&k)[E], uint32_t (&x)[E], int lane, uint32_t n) {
    if constexpr (KB <= LE + 6) {
        if (n > (1u << (KB - 1))) {  // (wave-uniform) below that the upper half of every block is padding: already in order
            if constexpr (KB <= LE) inlane_flip<E, KB>(k, x);
            else cross_flip<E, KB - LE>(k, x, lane);
            xor_steps<E, LE, KB - 2>(k, x, lane);
        }
        levels<E, LE, KB + 1>(k, x, lane, n);
    }
}

__device__ __forceinline__ int skew(int p) { return p + (p >> 5); }  // LDS index of element p: conflict-free lane-major AND position-major access

// One tile by one wave: sort, order ties, emit the sorted list and the four quadrant lists.
template <int E, int LE>
__device__ __forceinline__ void sort_tile_wave(const int tile, const uint32_t start, const uint32_t n, const uint32_t* __restrict__ u_depth,
                                               const uint32_t* __restrict__ u_val, const uint32_t* __restrict__ u_row,
                                               const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list,
                                               uint32_t* __restrict__ tile_keys, uint2* __restrict__ qlist, uint32_t* __restrict__ qrow,
                                               uint32_t* __restrict__ qcount, uint32_t* s_v, uint32_t* s_r) {
    const int lane = threadIdx.x;
    uint32_t key[E], idx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t p = (uint32_t)(lane * E + e);
        key[e] = p < n ? u_depth[start + p] : 0xFFFFFFFFu;  // (a real key is the bit pattern of a depth > 0.2: never all ones)
        idx[e] = p;
    }
    levels<E, LE, 1>(key, idx, lane, n);
    // ---- equal depths: Gaussian-id order (rasterizer_impl.cu:288-296 sorts id-ordered keys stably) ----
    bool tie = false;
#pragma unroll
    for (int e = 0; e + 1 < E; e++) tie |= (uint32_t)(lane * E + e + 1) < n && key[e] == key[e + 1];
    {
        const uint32_t nxt = (uint32_t)__shfl_down((int)key[0], 1, 64);
        tie |= lane < 63 && (uint32_t)(lane * E + E) < n && key[E - 1] == nxt;
    }
    if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
#pragma unroll
        for (int e = 0; e < E; e++) { s_v[skew(lane * E + e)] = key[e]; s_r[skew(lane * E + e)] = idx[e]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint32_t np[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t p = (uint32_t)(lane * E + e);
            np[e] = p;
            if (p < n) {
                const uint32_t k = key[e];
                uint32_t lo = p, hi = p;
                while (lo > 0u && s_v[skew((int)lo - 1)] == k) lo--;
                while (hi + 1u < n && s_v[skew((int)hi + 1)] == k) hi++;
                if (hi > lo) {
                    const uint32_t my = u_val[start + idx[e]] & GS2M_GID_MASK;
                    uint32_t rank = 0;
                    for (uint32_t q = lo; q <= hi; q++)
                        if (q != p) rank += (u_val[start + s_r[skew((int)q)]] & GS2M_GID_MASK) < my ? 1u : 0u;
                    np[e] = lo + rank;
                }
            }
        }
        // every position receives exactly one index (a permutation inside each run); one wave, in-order LDS: the reads above
        // are done before the first write below
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < E; e++) s_r[skew((int)np[e])] = idx[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < E; e++) idx[e] = s_r[skew(lane * E + e)];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- ids and rows of the sorted elements, lane-major -> LDS -> position-major ----
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t p = (uint32_t)(lane * E + e);
        uint32_t v = 0, r = 0;
        if (p < n) {
            v = u_val[start + idx[e]];
            r = u_row[start + idx[e]] + wave_rowbase[(v & GS2M_GID_MASK) >> 6];
        }
        s_v[skew((int)p)] = v;
        s_r[skew((int)p)] = r;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint2* out = qlist + (size_t)4 * start;
    uint32_t* orow = qrow + (size_t)4 * start;
    uint32_t run[4] = {0u, 0u, 0u, 0u};
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t base = 0; base < n; base += GS2M_WAVE) {
        const uint32_t k = base + (uint32_t)lane;
        uint32_t v = 0, r = 0;
        if (k < n) {
            v = s_v[skew((int)k)];
            r = s_r[skew((int)k)];
            point_list[start + k] = v;
            tile_keys[start + k] = (uint32_t)tile;
        }
        const uint32_t mask = v >> GS2M_GID_BITS;  // 0 for k >= n
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool hit = ((mask >> q) & 1u) != 0u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                const size_t o = (size_t)q * n + run[q] + (uint32_t)__popcll(m & lt);
                out[o] = make_uint2(v, k);
                orow[o] = r + (uint32_t)__popc(mask & ((1u << q) - 1u));
            }
            run[q] += (uint32_t)__popcll(m);
        }
    }
    if (lane < 4) qcount[tile * 4 + lane] = lane == 0 ? run[0] : (lane == 1 ? run[1] : (lane == 2 ? run[2] : run[3]));
}

template <int EMAX>  // 8: every span has at most 512 entries; 16: spans of up to 1024 entries are handled, longer ones left to the workgroup kernel
__global__ void __launch_bounds__(64) tile_sort_wave_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ u_depth,
                                                            const uint32_t* __restrict__ u_val, const uint32_t* __restrict__ u_row,
                                                            const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list,
                                                            uint32_t* __restrict__ tile_keys, uint2* __restrict__ qlist,
                                                            uint32_t* __restrict__ qrow, uint32_t* __restrict__ qcount) {
    constexpr int M = 64 * EMAX;
    __shared__ uint32_t s_v[M + M / 32], s_r[M + M / 32];
    const int tile = blockIdx.x;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    if (n == 0u) {
        if (threadIdx.x < 4) qcount[tile * 4 + threadIdx.x] = 0u;
        return;
    }
    if (n <= 512u) sort_tile_wave<8, 3>(tile, range.x, n, u_depth, u_val, u_row, wave_rowbase, point_list, tile_keys, qlist, qrow, qcount, s_v, s_r);
    else if constexpr (EMAX >= 16) {
        if (n <= 1024u) sort_tile_wave<16, 4>(tile, range.x, n, u_depth, u_val, u_row, wave_rowbase, point_list, tile_keys, qlist, qrow, qcount, s_v, s_r);
    }
}

// ---- long spans: a workgroup per tile ------------------------------------------------------------------------------------
constexpr int BIG_LDS = 4096;
__global__ void __launch_bounds__(256) tile_sort_big_kernel(const uint2* __restrict__ ranges, uint32_t* __restrict__ u_depth,
                                                            const uint32_t* __restrict__ u_val, const uint32_t* __restrict__ u_row,
                                                            const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list,
                                                            uint32_t* __restrict__ tile_keys, uint2* __restrict__ qlist,
                                                            uint32_t* __restrict__ qrow, uint32_t* __restrict__ qcount, uint32_t min_n) {
    __shared__ uint32_t s_key[BIG_LDS], s_idx[BIG_LDS], s_idx2[BIG_LDS];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x, start = range.x;
    if (n <= min_n) return;  // the wave kernel's
    // working arrays: LDS, or -- beyond its capacity -- the tile's own quadrant-list region (32 n bytes, written only at the end)
    const bool glob = n > (uint32_t)BIG_LDS;
    uint32_t* const K = glob ? reinterpret_cast<uint32_t*>(qlist + (size_t)4 * start) : s_key;
    uint32_t* I = glob ? K + n : s_idx;
    uint32_t* I2 = glob ? K + 2 * (size_t)n : s_idx2;
    auto barrier = [&]() {
        if (glob) __threadfence_block();
        gs2m_sync();
    };
    for (uint32_t p = tid; p < n; p += 256) { K[p] = u_depth[start + p]; I[p] = p; }
    barrier();
    int nlev = 0;
    while ((1u << nlev) < n) nlev++;
    auto cex = [&](uint32_t i, uint32_t p) {
        const uint32_t a = K[i], b = K[p];
        if (b < a) {
            K[i] = b; K[p] = a;
            const uint32_t t = I[i]; I[i] = I[p]; I[p] = t;
        }
    };
    for (int kb = 1; kb <= nlev; kb++) {
        const uint32_t half = 1u << (kb - 1), mask = (1u << kb) - 1u;
        for (uint32_t q = tid;; q += 256) {  // mirrored compare inside blocks of 2^kb; pairs with the partner in the (virtual, +inf) padding are no-ops
            const uint32_t i = ((q >> (kb - 1)) << kb) | (q & (half - 1u));
            if (i >= n) break;
            const uint32_t p = i ^ mask;
            if (p < n) cex(i, p);
        }
        barrier();
        for (int jb = kb - 2; jb >= 0; jb--) {
            const uint32_t j = 1u << jb;
            for (uint32_t q = tid;; q += 256) {
                const uint32_t i = ((q >> jb) << (jb + 1)) | (q & (j - 1u));
                if (i >= n) break;
                const uint32_t p = i | j;
                if (p < n) cex(i, p);
            }
            barrier();
        }
    }
    // equal depths -> id order
    bool tie = false;
    for (uint32_t p = tid; p + 1 < n; p += 256) tie |= K[p] == K[p + 1];
    if (gs2m_sync_or(tie)) {
        for (uint32_t p = tid; p < n; p += 256) {
            const uint32_t k = K[p];
            uint32_t lo = p, hi = p;
            while (lo > 0u && K[lo - 1] == k) lo--;
            while (hi + 1u < n && K[hi + 1] == k) hi++;
            uint32_t np = p;
            if (hi > lo) {
                const uint32_t my = u_val[start + I[p]] & GS2M_GID_MASK;
                uint32_t rank = 0;
                for (uint32_t q = lo; q <= hi; q++)
                    if (q != p) rank += (u_val[start + I[q]] & GS2M_GID_MASK) < my ? 1u : 0u;
                np = lo + rank;
            }
            I2[np] = I[p];
        }
        barrier();
        I = I2;
    }
    // sorted ids -> point_list (final), rows -> the span of u_depth (its keys were copied out above: dead)
    for (uint32_t p = tid; p < n; p += 256) {
        const uint32_t ix = I[p];
        const uint32_t v = u_val[start + ix];
        point_list[start + p] = v;
        tile_keys[start + p] = (uint32_t)tile;
        u_depth[start + p] = u_row[start + ix] + wave_rowbase[(v & GS2M_GID_MASK) >> 6];
    }
    __threadfence_block();
    gs2m_sync();
    // quadrant lists: wave q compacts quadrant q (the scratch in the list region is dead behind the barrier above)
    const int q = tid >> 6, lane = tid & 63;
    uint2* out = qlist + (size_t)4 * start + (size_t)q * n;
    uint32_t* orow = qrow + (size_t)4 * start + (size_t)q * n;
    uint32_t run = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t base = 0; base < n; base += GS2M_WAVE) {
        const uint32_t k = base + (uint32_t)lane;
        uint32_t v = 0, r = 0;
        if (k < n) { v = point_list[start + k]; r = u_depth[start + k]; }
        const uint32_t mask = v >> GS2M_GID_BITS;
        const bool hit = ((mask >> q) & 1u) != 0u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (hit) {
            const uint32_t o = run + (uint32_t)__popcll(m & lt);
            out[o] = make_uint2(v, k);
            orow[o] = r + (uint32_t)__popc(mask & ((1u << q) - 1u));
        }
        run += (uint32_t)__popcll(m);
    }
    if (lane == 0) qcount[tile * 4 + q] = run;
}

}  // namespace

void gs2m_launch_tile_sort(size_t tiles, uint32_t max_tile, const BinningState& b, const ImageState& im, const GeomState& g, hipStream_t s) {
    if (tiles == 0) return;
    if (max_tile <= 512u)
        tile_sort_wave_kernel<8><<<(unsigned)tiles, 64, 0, s>>>(im.ranges, b.u_depth, b.u_val, b.u_row, g.wave_rowbase, b.point_list, b.tile_keys, b.qlist, b.qrow, im.qcount);
    else
        tile_sort_wave_kernel<16><<<(unsigned)tiles, 64, 0, s>>>(im.ranges, b.u_depth, b.u_val, b.u_row, g.wave_rowbase, b.point_list, b.tile_keys, b.qlist, b.qrow, im.qcount);
    if (max_tile > 1024u)
        tile_sort_big_kernel<<<(unsigned)tiles, 256, 0, s>>>(im.ranges, b.u_depth, b.u_val, b.u_row, g.wave_rowbase, b.point_list, b.tile_keys, b.qlist, b.qrow, im.qcount, 1024u);
}
