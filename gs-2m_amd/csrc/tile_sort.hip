// Per-tile (depth, Gaussian id) sort on chip + the second binning level, for gfx950.
//
// binning.hip emits the instances in Gaussian-index order and the stable radix sort by tile id (radix_sort.hip) leaves every
// tile's span in that order.  This file gives every span the reference's order -- ascending view depth, ties in Gaussian-id
// order: what its 45-bit radix sort of (tile << 32 | depth) keys over the id-ordered duplicateWithKeys output produces
// (rasterizer_impl.cu:288-296) -- writes ranges[] (identifyTileRanges: the sort's last pass recorded where
// every tile's run starts and ends) and splits the sorted list into the four per-quadrant lists the blend kernels walk.
//   * spans of up to 1024 entries (every tile of the bench scenes): ONE WAVE per tile, the span in registers as (depth, position
//     in the span) pairs, lane-major (lane l holds elements l E .. l E + E - 1, E = 8 or 16), sorted by a bitonic network in its
//     all-ascending form (each merge starts with a mirrored compare, so that padding with +inf needs no direction bits):
//     strides inside a lane are register renaming + compare-exchange, strides across lanes are DPP moves (quad_perm,
//     row_half_mirror / row_mirror, row_ror:8, a masked row_shl:4 / row_shr:4 pair) and, for the three strides that cross a
//     DPP row, ds_bpermute; levels above the span's length are skipped (their input is already in order);
//   * equal depths (two Gaussians at exactly the same view depth): the network compares depths only; when the sorted span holds
//     equal neighbours (rare, wave-uniform) it is run again with (depth, position) as the key -- positions are in id order;
//   * ids and rows wait in LDS under their span position and are picked up in sorted order, then ballots give every instance
//     its place in each of the four quadrant lists with coalesced stores; the gradient row of a list entry = the instance's
//     first row (relative to its emit wave, emit_kernel) + the wave's base (rowscan_kernel) + its quadrants before this one;
//   * longer spans are queued and sorted by a workgroup each: the same network over LDS (up to 4096 entries) or, beyond, over
//     the tile's own (still unused) quadrant-list region in global memory -- slow, correct, exercised by the dense-scene tests.
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
template <int E, bool TIE>  // TIE: equal keys are ordered by x (the span position)
__device__ __forceinline__ void ce(uint32_t (&k)[E], uint32_t (&x)[E], int a, int b) {  // a < b: the smaller key to a
    const bool sw = TIE ? (k[b] < k[a] || (k[b] == k[a] && x[b] < x[a])) : (k[b] < k[a]);
    const uint32_t ka = sw ? k[b] : k[a], kb = sw ? k[a] : k[b], xa = sw ? x[b] : x[a], xb = sw ? x[a] : x[b];
    k[a] = ka; k[b] = kb; x[a] = xa; x[b] = xb;
}
template <int E, int KB, bool TIE>  // mirrored compare inside blocks of 2^KB elements of one lane
__device__ __forceinline__ void inlane_flip(uint32_t (&k)[E], uint32_t (&x)[E]) {
#pragma unroll
    for (int e = 0; e < E; e++)
        if (((e >> (KB - 1)) & 1) == 0) ce<E, TIE>(k, x, e, e ^ ((1 << KB) - 1));
}
template <int E, int JB, bool TIE>
__device__ __forceinline__ void inlane_xor(uint32_t (&k)[E], uint32_t (&x)[E]) {
#pragma unroll
    for (int e = 0; e < E; e++)
        if (((e >> JB) & 1) == 0) ce<E, TIE>(k, x, e, e | (1 << JB));
}
// Across lanes: the lane with the lower number keeps the smaller keys.  On equal keys both keep their own (consistent on
// both sides; the order of equal depths is settled by a second run with TIE).
template <int E, int T, bool TIE>
__device__ __forceinline__ void cross_flip(uint32_t (&k)[E], uint32_t (&x)[E], int lane) {
    const bool lower = ((lane >> (T - 1)) & 1) == 0;
    uint32_t nk[E], nx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t ok = lane_flip<T>(k[E - 1 - e], lane), ox = lane_flip<T>(x[E - 1 - e], lane);
        const bool take = TIE ? (lower ? (ok < k[e] || (ok == k[e] && ox < x[e])) : (ok > k[e] || (ok == k[e] && ox > x[e])))
                              : (lower ? (ok < k[e]) : (ok > k[e]));
        nk[e] = take ? ok : k[e];
        nx[e] = take ? ox : x[e];
    }
#pragma unroll
    for (int e = 0; e < E; e++) { k[e] = nk[e]; x[e] = nx[e]; }
}
template <int E, int B, bool TIE>
__device__ __forceinline__ void cross_xor(uint32_t (&k)[E], uint32_t (&x)[E], int lane) {
    const bool lower = ((lane >> B) & 1) == 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t ok = lane_xor<B>(k[e], lane), ox = lane_xor<B>(x[e], lane);
        const bool take = TIE ? (lower ? (ok < k[e] || (ok == k[e] && ox < x[e])) : (ok > k[e] || (ok == k[e] && ox > x[e])))
                              : (lower ? (ok < k[e]) : (ok > k[e]));
        k[e] = take ? ok : k[e];
        x[e] = take ? ox : x[e];
    }
}
template <int E, int LE, int JB, bool TIE>  // compare-exchange steps with strides 2^JB ... 2^0
__device__ __forceinline__ void xor_steps(uint32_t (&k)[E], uint32_t (&x)[E], int lane) {
    if constexpr (JB >= 0) {
        if constexpr (JB >= LE) cross_xor<E, JB - LE, TIE>(k, x, lane);
        else inlane_xor<E, JB, TIE>(k, x);
        xor_steps<E, LE, JB - 1, TIE>(k, x, lane);
    }
}
template <int E, int LE, int KB, bool TIE>  // merge levels KB ... LE + 6 (sorted blocks of 2^(KB-1) -> 2^KB); a level whose blocks are longer than the data is a no-op
__device__ __forceinline__ void levels(uint32_t (&k)[E], uint32_t (&x)[E], int lane, uint32_t n) {
    if constexpr (KB <= LE + 6) {
        if (n > (1u << (KB - 1))) {  // (wave-uniform) below that the upper half of every block is padding: already in order
            if constexpr (KB <= LE) inlane_flip<E, KB, TIE>(k, x);
            else cross_flip<E, KB - LE, TIE>(k, x, lane);
            xor_steps<E, LE, KB - 2, TIE>(k, x, lane);
        }
        levels<E, LE, KB + 1, TIE>(k, x, lane, n);
    }
}

__device__ __forceinline__ int skew(int p) { return p + (p >> 5); }  // LDS index of element p: conflict-free lane-major AND position-major access

// One tile by one wave: sort, order ties, emit the sorted list and the four quadrant lists.
template <int E, int LE>
__device__ __forceinline__ void sort_tile_wave(const int tile, const uint32_t start, const uint32_t n, const uint32_t* __restrict__ slot_sorted,
                                               const uint2* __restrict__ e_vr, const uint32_t* __restrict__ depth_key,
                                               const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list,
                                               uint2* __restrict__ qlist, uint32_t* __restrict__ qrow,
                                               uint32_t* __restrict__ qcount, uint32_t* s_v, uint32_t* s_r) {
    const int lane = threadIdx.x;
    uint32_t key[E], idx[E];
    {
        uint32_t slot[E];
        uint2 vr[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t p = (uint32_t)(lane * E + e);
            slot[e] = p < n ? slot_sorted[start + p] : 0u;
        }
#pragma unroll
        for (int e = 0; e < E; e++) vr[e] = (uint32_t)(lane * E + e) < n ? e_vr[slot[e]] : make_uint2(0u, 0u);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t p = (uint32_t)(lane * E + e);
            const uint32_t gid = vr[e].x & GS2M_GID_MASK;
            key[e] = p < n ? depth_key[gid] : 0xFFFFFFFFu;  // (a real key is the bit pattern of a depth > 0.2: never all ones)
            idx[e] = p;
            // id | mask and the absolute first row wait in LDS under their span position
            s_v[skew((int)p)] = vr[e].x;
            s_r[skew((int)p)] = p < n ? vr[e].y + wave_rowbase[gid >> 6] : 0u;
        }
    }
    levels<E, LE, 1, false>(key, idx, lane, n);
    // ---- equal depths: span order = Gaussian-id order (rasterizer_impl.cu:288-296 sorts id-ordered keys stably) ----
    bool tie = false;
#pragma unroll
    for (int e = 0; e + 1 < E; e++) tie |= (uint32_t)(lane * E + e + 1) < n && key[e] == key[e + 1];
    {
        const uint32_t nxt = (uint32_t)__shfl_down((int)key[0], 1, 64);
        tie |= lane < 63 && (uint32_t)(lane * E + E) < n && key[E - 1] == nxt;
    }
    if (__builtin_amdgcn_ballot_w64(tie) != 0ull) levels<E, LE, 1, true>(key, idx, lane, n);  // (depth, position): a total order
    // ---- ids and rows of the sorted elements: picked up by position, parked again lane-major, read position-major ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t sv[E], sr[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const bool real = (uint32_t)(lane * E + e) < n;  // (padding sorts behind every real element: idx < n for the first n)
        sv[e] = real ? s_v[skew((int)idx[e])] : 0u;
        sr[e] = real ? s_r[skew((int)idx[e])] : 0u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int e = 0; e < E; e++) {
        s_v[skew(lane * E + e)] = sv[e];
        s_r[skew(lane * E + e)] = sr[e];
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

__global__ void __launch_bounds__(64) tile_sort_wave_kernel(const uint32_t* __restrict__ ranges_raw, uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ slot_sorted, const uint2* __restrict__ e_vr,
                                                            const uint32_t* __restrict__ depth_key, const uint32_t* __restrict__ wave_rowbase,
                                                            uint32_t* __restrict__ point_list, uint2* __restrict__ qlist,
                                                            uint32_t* __restrict__ qrow, uint32_t* __restrict__ qcount, uint32_t* bigq) {
    constexpr int M = 64 * 16;
    __shared__ uint32_t s_v[M + M / 32], s_r[M + M / 32];
    const int tile = blockIdx.x;
    // identifyTileRanges (rasterizer_impl.cu:108-129): the tile sort's last pass recorded where the tile's run of instances starts
    // and ends (radix_sort.hip: range_raw); (0, 0) for an untouched tile, as the reference's memset leaves it
    const uint2 raw = reinterpret_cast<const uint2*>(ranges_raw)[tile];
    const uint2 range = raw.y != 0u ? make_uint2(~raw.x, raw.y) : make_uint2(0u, 0u);
    if (threadIdx.x == 0) ranges[tile] = range;
    const uint32_t n = range.y - range.x;
    if (n == 0u) {
        if (threadIdx.x < 4) qcount[tile * 4 + threadIdx.x] = 0u;
        return;
    }
    if (n <= 512u) sort_tile_wave<8, 3>(tile, range.x, n, slot_sorted, e_vr, depth_key, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r);
    else if (n <= 1024u) sort_tile_wave<16, 4>(tile, range.x, n, slot_sorted, e_vr, depth_key, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r);
    else if (threadIdx.x == 0) bigq[1u + atomicAdd(&bigq[0], 1u)] = (uint32_t)tile;  // a workgroup's (tile_sort_big_kernel)
}

// ---- long spans: a workgroup per queued tile ---------------------------------------------------------------------------------
constexpr int BIG_LDS = 4096;
__global__ void __launch_bounds__(256) tile_sort_big_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ slot_sorted,
                                                            const uint2* __restrict__ e_vr, const uint32_t* __restrict__ depth_key,
                                                            const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list,
                                                            uint32_t* __restrict__ row_tmp /* R words: sorted rows on their way to the lists */,
                                                            uint2* __restrict__ qlist, uint32_t* __restrict__ qrow,
                                                            uint32_t* __restrict__ qcount, const uint32_t* __restrict__ bigq) {
    __shared__ uint32_t s_key[BIG_LDS], s_idx[BIG_LDS];
    const int tid = threadIdx.x;
    const uint32_t nbig = bigq[0];
    for (uint32_t bq = blockIdx.x; bq < nbig; bq += gridDim.x) {
        const int tile = (int)bigq[1u + bq];
        const uint2 range = ranges[tile];
        const uint32_t n = range.y - range.x, start = range.x;
        // working arrays: LDS, or -- beyond its capacity -- the tile's own quadrant-list region (32 n bytes, written only at the end)
        const bool glob = n > (uint32_t)BIG_LDS;
        uint32_t* const K = glob ? reinterpret_cast<uint32_t*>(qlist + (size_t)4 * start) : s_key;
        uint32_t* const I = glob ? K + n : s_idx;
        auto barrier = [&]() {
            if (glob) __threadfence_block();
            gs2m_sync();
        };
        barrier();  // the previous tile's LDS reads are done
        for (uint32_t p = tid; p < n; p += 256) {
            K[p] = depth_key[e_vr[slot_sorted[start + p]].x & GS2M_GID_MASK];
            I[p] = p;
        }
        barrier();
        int nlev = 0;
        while ((1u << nlev) < n) nlev++;
        auto cex = [&](uint32_t i, uint32_t p) {  // (depth, span position): a total order; positions are in Gaussian-id order
            const uint32_t a = K[i], b = K[p], xa = I[i], xb = I[p];
            if (b < a || (b == a && xb < xa)) { K[i] = b; K[p] = a; I[i] = xb; I[p] = xa; }
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
        // sorted ids -> point_list (final), absolute rows -> row_tmp
        for (uint32_t p = tid; p < n; p += 256) {
            const uint2 vr = e_vr[slot_sorted[start + I[p]]];
            point_list[start + p] = vr.x;
            row_tmp[start + p] = vr.y + wave_rowbase[(vr.x & GS2M_GID_MASK) >> 6];
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
            if (k < n) { v = point_list[start + k]; r = row_tmp[start + k]; }
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
}

}  // namespace

void gs2m_launch_tile_sort(size_t tiles, const BinningState& b, const ImageState& im, const GeomState& g, hipStream_t s) {
    if (tiles == 0) return;
    tile_sort_wave_kernel<<<(unsigned)tiles, 64, 0, s>>>(im.ranges_raw, im.ranges, b.slot_sorted, b.e_vr, g.depth_key, g.wave_rowbase, b.point_list,
                                                         b.qlist, b.qrow, im.qcount, im.bigq);
    // tiles of more than 1024 instances were queued (none on the bench scenes: the workgroups find an empty queue)
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    tile_sort_big_kernel<<<grid, 256, 0, s>>>(im.ranges, b.slot_sorted, b.e_vr, g.depth_key, g.wave_rowbase, b.point_list, b.sort_valA, b.qlist, b.qrow,
                                              im.qcount, im.bigq);
}
