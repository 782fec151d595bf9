// Per-tile (depth, Gaussian id) sort on chip + the second binning level, for gfx950.
//
// binning.hip emits the instances in Gaussian-index order and the stable radix sort by tile id (radix_sort.hip) leaves every
// tile's span in that order.  This file gives every span the reference's order -- ascending view depth, ties in Gaussian-id
// order: what its 45-bit radix sort of (tile << 32 | depth) keys over the id-ordered duplicateWithKeys output produces
// (rasterizer_impl.cu:288-296) -- writes ranges[] (identifyTileRanges: the sort's last pass recorded where
// every tile's run starts and ends) and splits the sorted list into the four per-quadrant lists the blend kernels walk.
//   * the span in registers as (depth, position in the span) pairs, lane-major (lane l holds elements l E .. l E + E - 1), sorted
//     by a bitonic network in its all-ascending form (each merge starts with a mirrored compare, so that padding with +inf needs no
//     direction bits): strides inside a lane are register renaming + compare-exchange, strides across lanes are DPP moves (quad_perm,
//     row_half_mirror / row_mirror, row_ror:8, a masked row_shl:4 / row_shr:4 pair) and, for the strides that cross a DPP row,
//     ds_bpermute; strides across the waves of a workgroup go through LDS;
//   * who sorts which span (gs2m_launch_tile_sort below): ONE WAVE per tile (8 or 16 elements per lane: spans of up to 512 / 1024
//     entries) on frames of many tiles, a WORKGROUP per tile (1 to 16 elements per lane: up to 4096 entries) on frames of few tiles and
//     for long spans everywhere; beyond 4096 entries sorted runs of 4096 from the same network, merged over global memory;
//   * equal depths (coincident Gaussians: a quarter of a trained model's tiles hold such a pair): the network compares depths only;
//     the members of a run of equal depths then look along the run for their place in span order = id order (tie_positions);
//   * ids and rows wait in LDS under their span position and are picked up in sorted order, then ballots give every instance
//     its place in each of the four quadrant lists with coalesced stores; the gradient row of a list entry = the instance's
//     first row (relative to its emit wave, emit_kernel; absolute and flagged GS2M_ROWS_BIG for a heavy Gaussian's) + the wave's base
//     (rowscan_kernel) + its quadrants before this one.
#include "common.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

// ---- cross-lane moves ---------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
// a full permutation inside the row: every lane has a source, `old` is never used (bound_ctrl: no tied operand, no copy)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// value of lane ^ (1 << B)
template <int B>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
    if constexpr (B == 0) return dpp_perm<DPP_QUAD_PERM(1, 0, 3, 2)>(v);
    else if constexpr (B == 1) return dpp_perm<DPP_QUAD_PERM(2, 3, 0, 1)>(v);
    else if constexpr (B == 2) {
        // banks (groups of 4 lanes) 0 and 2 of every row read 4 lanes up, banks 1 and 3 read 4 lanes down
        const uint32_t t = dpp_u32<0x104 /* row_shl:4 */, 0xF, 0x5>(v, v);
        return dpp_u32<0x114 /* row_shr:4 */, 0xF, 0xA>(t, v);
    } else if constexpr (B == 3) return dpp_perm<0x128 /* row_ror:8 */>(v);
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ (1 << B)) << 2, (int)v);
}
// value of lane ^ ((1 << T) - 1): mirrored inside groups of 2^T lanes
template <int T>
__device__ __forceinline__ uint32_t lane_flip(uint32_t v, int lane) {
    if constexpr (T == 1) return dpp_perm<DPP_QUAD_PERM(1, 0, 3, 2)>(v);
    else if constexpr (T == 2) return dpp_perm<DPP_QUAD_PERM(3, 2, 1, 0)>(v);
    else if constexpr (T == 3) return dpp_perm<DPP_ROW_HALF_MIRROR>(v);
    else if constexpr (T == 4) return dpp_perm<DPP_ROW_MIRROR>(v);
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ ((1 << T) - 1)) << 2, (int)v);
}

// ---- the network, element i = lane * E + e --------------------------------------------------------------------------------
// compile-time loop: register-array indices must be constants (a run-time index becomes a chain of compares and selects)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int E, int a, int b>
__device__ __forceinline__ void ce(uint32_t (&k)[E], uint32_t (&x)[E]) {  // a < b: the smaller key to a
    // min / max for the keys, one compare for the two payload selects: 5 vector instructions, no mask arithmetic
    const bool sw = k[b] < k[a];
    const uint32_t mn = min(k[a], k[b]), mx = max(k[a], k[b]);
    const uint32_t xa = sw ? x[b] : x[a], xb = sw ? x[a] : x[b];
    k[a] = mn; k[b] = mx; x[a] = xa; x[b] = xb;
}
template <int E, int KB>  // mirrored compare inside blocks of 2^KB elements of one lane
__device__ __forceinline__ void inlane_flip(uint32_t (&k)[E], uint32_t (&x)[E]) {
    static_for<0, E>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if constexpr (((e >> (KB - 1)) & 1) == 0) ce<E, e, (e ^ ((1 << KB) - 1))>(k, x);
    });
}
template <int E, int JB>
__device__ __forceinline__ void inlane_xor(uint32_t (&k)[E], uint32_t (&x)[E]) {
    static_for<0, E>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if constexpr (((e >> JB) & 1) == 0) ce<E, e, (e | (1 << JB))>(k, x);
    });
}
// Across lanes: the lane with the lower number keeps the smaller keys.  On equal keys both keep their own (consistent on
// both sides; runs of equal depths are put into span order afterwards: tie_positions).
template <int E, int T>
__device__ __forceinline__ void cross_flip(uint32_t (&k)[E], uint32_t (&x)[E], int lane) {
    const bool lower = ((lane >> (T - 1)) & 1) == 0;
    uint32_t nk[E], nx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t ok = lane_flip<T>(k[E - 1 - e], lane), ox = lane_flip<T>(x[E - 1 - e], lane);
        // the lower lane keeps the minimum, the upper one the maximum; the payload follows the key (equal keys: each keeps its own)
        const uint32_t mn = min(k[e], ok), mx = max(k[e], ok);
        nk[e] = lower ? mn : mx;
        nx[e] = nk[e] == k[e] ? x[e] : ox;
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
        const uint32_t mn = min(k[e], ok), mx = max(k[e], ok);
        const uint32_t nk = lower ? mn : mx;
        x[e] = nk == k[e] ? x[e] : ox;
        k[e] = nk;
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

// ---- equal depths ------------------------------------------------------------------------------------------------------------------
// The network orders by depth alone; the members of a run of equal depths must follow in span order (= Gaussian-id order: what the
// reference's stable radix sort of id-ordered keys leaves, rasterizer_impl.cu:288-296).  Trained models hold coincident Gaussians --
// a quarter of the tiles of a 420 k-Gaussian training view have such a pair (tools/c4_ts_clock.py) -- so the runs are settled where
// they lie: with keys and span positions parked in sorted order (s_k, s_i), element i looks along its run and goes to
// run start + the number of members with a smaller span position.  Linear in the run's length per member: coincident Gaussians come in
// pairs; a tile whose instances ALL share one depth (the tests build such spans) still comes out right, in ~n^2 / lanes steps.
template <int E>
__device__ __forceinline__ void tie_positions(const uint32_t (&key)[E], const uint32_t (&idx)[E], uint32_t (&pos)[E], const int t, const uint32_t n,
                                              const uint32_t* s_k, const uint32_t* s_i) {
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t i = (uint32_t)(t * E + e);
        pos[e] = i;
        if (i >= n) continue;
        const uint32_t K = key[e], X = idx[e];
        uint32_t a = i, cnt = 0;
        for (uint32_t j = i; j > 0u;) {
            j--;
            if (s_k[skew((int)j)] != K) break;
            a = j;
            cnt += s_i[skew((int)j)] < X ? 1u : 0u;
        }
        for (uint32_t j = i + 1u; j < n; j++) {
            if (s_k[skew((int)j)] != K) break;
            cnt += s_i[skew((int)j)] < X ? 1u : 0u;
        }
        pos[e] = a + cnt;
    }
}

// One tile by one wave: the span's emission slots (coalesced: span position p = e * 64 + lane sits in register e of lane `lane` --
// the network sorts whatever arrangement it is given) and, through them, the 16-byte records (one gather each, 8 in flight per
// lane); staging in LDS, the network, ties, the sorted list and the four quadrant lists.
template <int E, int LE>
__device__ __forceinline__ void sort_tile_wave(const int tile, const uint32_t start, const uint32_t n, const uint32_t* __restrict__ slot_sorted,
                                               const uint4* __restrict__ e_rec, const uint32_t* __restrict__ wave_rowbase,
                                               uint32_t* __restrict__ point_list, uint2* __restrict__ qlist, uint32_t* __restrict__ qrow,
                                               uint32_t* __restrict__ qcount, uint32_t* s_v, uint32_t* s_r, const int lane) {
    uint32_t key[E], idx[E], rb[E];  // rb: first row of the emit wave that owns the element at span position e * 64 + lane
    static_for<0, E / 8>([&](auto bc) {  // batches of 8 elements per lane: 32 registers of records in flight, whatever E
        constexpr int e0 = 8 * decltype(bc)::value;
        uint32_t slot[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t p = (uint32_t)((e0 + e) * GS2M_WAVE + lane);
            slot[e] = p < n ? slot_sorted[start + p] : 0u;
        }
        uint4 rc[8];  // {id | mask, relative first row, depth key, -}
#pragma unroll
        for (int e = 0; e < 8; e++) rc[e] = (uint32_t)((e0 + e) * GS2M_WAVE + lane) < n ? e_rec[slot[e]] : make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
        static_for<0, 8>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const uint32_t p = (uint32_t)((e0 + e) * GS2M_WAVE + lane);
            key[e0 + e] = rc[e].z;  // (a real key is the bit pattern of a depth > 0.2: never all ones, the padding's key)
            idx[e0 + e] = p;
            // id | mask and the first row wait in LDS under their span position; the row's base (a dependent gather) is asked
            // for now and added behind the sort: its latency disappears behind the network
            s_v[skew((int)p)] = rc[e].x;
            s_r[skew((int)p)] = rc[e].y & ~GS2M_ROWS_BIG;
            rb[e0 + e] = p < n && (rc[e].y & GS2M_ROWS_BIG) == 0u ? wave_rowbase[(rc[e].x & GS2M_GID_MASK) >> 6] : 0u;  // (a heavy instance's row is absolute)
        });
        asm volatile("" ::: "memory");  // the next batch's loads stay behind this batch's staging (registers: one batch in flight)
    });
    // (the padding is spread over the lanes in this arrangement: every level runs)
    levels<E, LE, 1>(key, idx, lane, 0xFFFFFFFFu);
#pragma unroll
    for (int e = 0; e < E; e++) s_r[skew(e * GS2M_WAVE + lane)] += rb[e];
    // ---- ids and rows of the sorted elements: picked up by span position (sorted element i = lane * E + e, the network's index space) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t sv[E], sr[E];
    auto pick_up = [&]() {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool real = (uint32_t)(lane * E + e) < n;  // (padding sorts behind every real element: idx < n for the first n)
            sv[e] = real ? s_v[skew((int)idx[e])] : 0u;
            sr[e] = real ? s_r[skew((int)idx[e])] : 0u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto park = [&]() {  // parked again under their sorted position, read position-major below
#pragma unroll
        for (int e = 0; e < E; e++) {
            s_v[skew(lane * E + e)] = sv[e];
            s_r[skew(lane * E + e)] = sr[e];
        }
    };
    pick_up();
    // ---- equal depths (tie_positions above) ----
    bool tie = false;
#pragma unroll
    for (int e = 0; e + 1 < E; e++) tie |= (uint32_t)(lane * E + e + 1) < n && key[e] == key[e + 1];
    {
        const uint32_t nxt = (uint32_t)__shfl_down((int)key[0], 1, 64);
        tie |= lane < 63 && (uint32_t)(lane * E + E) < n && key[E - 1] == nxt;
    }
    if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {  // (wave-uniform)
#pragma unroll
        for (int e = 0; e < E; e++) {  // (the staging arrays are free: their content is in sv / sr)
            s_v[skew(lane * E + e)] = key[e];
            s_r[skew(lane * E + e)] = idx[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint32_t pos[E];
        tie_positions<E>(key, idx, pos, lane, n, s_v, s_r);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < E; e++) {
            s_v[skew((int)pos[e])] = sv[e];
            s_r[skew((int)pos[e])] = sr[e];
        }
    } else {
        park();
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS arrays are the next tile's from here on
}

// Which tile a one-wave workgroup takes: workgroup ids go round the 8 XCDs (each with its own L2), and the records a tile gathers are
// 16 bytes out of a 64-byte sector whose other three records usually belong to the SAME Gaussian's instances in the neighbouring
// tiles (emit_kernel writes a Gaussian's instances to consecutive slots).  So the four tiles of a 2 x 2 block run back to back on
// ONE XCD: the sector one of them pulls in is in that L2 when the others ask for it (4 x 4 tiles since the measurement below).  -1: no tile (grid padding).
#ifndef GS2M_TS_SUPER_LOG
#define GS2M_TS_SUPER_LOG 2  // the block of tiles that shares an XCD is 2^LOG x 2^LOG tiles (measured: 1x1 82 us, 2x2 80, 4x4 73, 8x8 72)
#endif
__device__ __forceinline__ int tile_of_block(int b, int tiles_x, int tiles_y) {
    constexpr int L = GS2M_TS_SUPER_LOG, SIDE = 1 << L, PER = SIDE * SIDE;
    const int xcd = b & 7, j = b >> 3, sub = j & (PER - 1);
    const int stx = (tiles_x + SIDE - 1) >> L, nsuper = stx * ((tiles_y + SIDE - 1) >> L);
    const int S = (j >> (2 * L)) * 8 + xcd;
    if (S >= nsuper) return -1;
    const int tx = SIDE * (S % stx) + (sub & (SIDE - 1)), ty = SIDE * (S / stx) + (sub >> L);
    if (tx >= tiles_x || ty >= tiles_y) return -1;
    return ty * tiles_x + tx;
}
__host__ __device__ inline unsigned tile_grid(int tiles_x, int tiles_y) {
    constexpr int L = GS2M_TS_SUPER_LOG, SIDE = 1 << L;
    const size_t nsuper = (size_t)((tiles_x + SIDE - 1) >> L) * ((tiles_y + SIDE - 1) >> L);
    return (unsigned)(((nsuper + 7) / 8) * 8 * SIDE * SIDE);
}

// identifyTileRanges (rasterizer_impl.cu:108-129): the tile sort's last pass recorded where the tile's run of instances starts
// and ends (radix_sort.hip: range_raw); (0, 0) for an untouched tile, as the reference's memset leaves it
__device__ __forceinline__ uint2 tile_range(const uint32_t* __restrict__ ranges_raw, uint2* __restrict__ ranges, int tile, int lane) {
    const uint2 raw = reinterpret_cast<const uint2*>(ranges_raw)[tile];
    const uint2 range = raw.y != 0u ? make_uint2(~raw.x, raw.y) : make_uint2(0u, 0u);
    if (lane == 0) ranges[tile] = range;
    return range;
}

// One wave per tile; spans of up to 512 entries are sorted here (8 elements per lane: 4 KB of LDS and ~45 registers per wave, i.e.
// a full complement of waves per SIMD), longer ones are left to the two kernels below, which look at every tile's length themselves
// (a queue of tile ids filled with one atomic per long tile cost 97 us at 2 M Gaussians, where EVERY tile is long: 8160 atomics on
// one word).  (Two or three tiles per wave with the later tiles' records in flight while the first is sorted: 127 / 122 us against
// 83 -- the extra registers cost more occupancy than the overlap returns.)
__global__ void __launch_bounds__(64) tile_sort_wave_kernel(const uint32_t* __restrict__ ranges_raw, uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ slot_sorted, const uint4* __restrict__ e_rec,
                                                            const uint32_t* __restrict__ wave_rowbase,
                                                            uint32_t* __restrict__ point_list, uint2* __restrict__ qlist,
                                                            uint32_t* __restrict__ qrow, uint32_t* __restrict__ qcount, int tiles_x, int tiles_y,
                                                            uint32_t wave_max /* 512: longer spans are another kernel's */, uint32_t* __restrict__ counters) {
    constexpr int M = 64 * 8;
    __shared__ uint32_t s_v[M + M / 32], s_r[M + M / 32];
    const int lane = threadIdx.x;
    const int tile = tile_of_block(blockIdx.x, tiles_x, tiles_y);
    if (tile < 0) return;
    const uint2 range = tile_range(ranges_raw, ranges, tile, lane);
    const uint32_t n = range.y - range.x;
    if (n == 0u) {
        if (lane < 4) qcount[tile * 4 + lane] = 0u;
    } else if (n <= wave_max) {
        sort_tile_wave<8, 3>(tile, range.x, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, lane);
    } else if (lane == 0) {  // tells the kernels behind this one that they have work (a plain store of 1: any number of writers)
        counters[n <= 1024u ? GS2M_CNT_SPAN_MID : GS2M_CNT_SPAN_LONG] = 1u;
    }
}

// ---- longer spans ---------------------------------------------------------------------------------------------------------------
// 513 .. 1024 entries: one wave, 16 elements per lane (8 KB of LDS, 128 registers).  Launched over all tiles: a wave whose tile is
// shorter or longer leaves at once (on the bench scene all of them; at 2 M Gaussians / 1080p every tile is sorted here).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
tile_sort_wave16_kernel(const uint32_t* __restrict__ ranges_raw, const uint32_t* __restrict__ slot_sorted, const uint4* __restrict__ e_rec,
                        const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list, uint2* __restrict__ qlist,
                        uint32_t* __restrict__ qrow, uint32_t* __restrict__ qcount, int tiles_x, int tiles_y, const uint32_t* __restrict__ counters) {
    constexpr int M16 = 64 * 16, W16 = M16 + M16 / 32;
    __shared__ uint32_t s_v[W16], s_r[W16];
    if (counters[GS2M_CNT_SPAN_MID] == 0u) return;  // no such span in this frame (tile_sort_wave_kernel looked)
    const int lane = threadIdx.x;
    const int tile = tile_of_block(blockIdx.x, tiles_x, tiles_y);
    if (tile < 0) return;
    const uint2 raw = reinterpret_cast<const uint2*>(ranges_raw)[tile];
    const uint32_t start = ~raw.x, n = raw.y != 0u ? raw.y - start : 0u;
    if (n <= 512u || n > 1024u) return;
    sort_tile_wave<16, 4>(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, lane);
}

// ---- a workgroup per tile, the span still in registers -------------------------------------------------------------------------------
// Few tiles with long lists (a 777 x 581 training view: 1813 tiles of ~500 instances; 2 M Gaussians at 1080p: 8160 tiles of ~660) leave
// a wave-per-tile kernel one or two waves per SIMD, each a serial chain of 16 gathers per lane and a 1024-element network.  Here
// 256 lanes share a tile: element i = tid * E + e (E = 2, 4 or 8: up to 512, 1024, 2048 entries), every wave sorts its 64 E elements
// with the network above, and the two merge levels that span the four waves exchange registers through LDS -- three exchanges in
// all (the mirrored compare of either level and the stride-of-one-wave step of the last), everything else stays inside a wave.
template <int E>
__device__ __forceinline__ void cross_wave(uint32_t (&k)[E], uint32_t (&x)[E], const int tid, const int partner, const bool mirrored, const bool lower,
                                           uint32_t* s_x) {  // one exchange array: the keys go across first, then the positions
    uint32_t ok[E], ox[E];
#pragma unroll
    for (int e = 0; e < E; e++) s_x[skew(tid * E + e)] = k[e];
    gs2m_sync();
#pragma unroll
    for (int e = 0; e < E; e++) ok[e] = s_x[skew(partner * E + (mirrored ? E - 1 - e : e))];
    gs2m_sync();
#pragma unroll
    for (int e = 0; e < E; e++) s_x[skew(tid * E + e)] = x[e];
    gs2m_sync();
#pragma unroll
    for (int e = 0; e < E; e++) ox[e] = s_x[skew(partner * E + (mirrored ? E - 1 - e : e))];
    gs2m_sync();  // the exchange array is rewritten by the next exchange
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t mn = min(k[e], ok[e]), mx = max(k[e], ok[e]);
        const uint32_t nk = lower ? mn : mx;
        x[e] = nk == k[e] ? x[e] : ox[e];
        k[e] = nk;
    }
}
template <int E, int LE>
__device__ __forceinline__ void network_wg(uint32_t (&k)[E], uint32_t (&x)[E], const int tid, uint32_t* s_x) {
    const int lane = tid & 63, wave = tid >> 6;
    levels<E, LE, 1>(k, x, lane, 0xFFFFFFFFu);                                        // sorted runs of 64 E: one per wave
    cross_wave<E>(k, x, tid, tid ^ 127, true, (wave & 1) == 0, s_x);            // level LE + 7: mirrored compare across a pair of waves
    xor_steps<E, LE, LE + 5>(k, x, lane);
    cross_wave<E>(k, x, tid, tid ^ 255, true, wave < 2, s_x);                   // level LE + 8: across all four
    cross_wave<E>(k, x, tid, tid ^ 64, false, (wave & 1) == 0, s_x);            //   stride of one wave
    xor_steps<E, LE, LE + 5>(k, x, lane);
}

constexpr int WG_MAX = 4096;
template <int E, int LE>
__device__ __forceinline__ void sort_tile_wg(const int tile, const uint32_t start, const uint32_t n, const uint32_t* __restrict__ slot_sorted,
                                             const uint4* __restrict__ e_rec, const uint32_t* __restrict__ wave_rowbase,
                                             uint32_t* __restrict__ point_list, uint2* __restrict__ qlist, uint32_t* __restrict__ qrow,
                                             uint32_t* __restrict__ qcount, uint32_t* s_v, uint32_t* s_r, uint32_t* s_x, const int tid,
                                             uint32_t* __restrict__ clk = nullptr) {
    const int lane = tid & 63, wave = tid >> 6;
#ifdef GS2M_TS_CLOCK  // variant builds: where a tile's time goes (100 MHz wall clock ticks of thread 0 at the phase boundaries)
    const unsigned long long c0 = wall_clock64();
    int cphase = 0;
#define TS_CLK() do { if (tid == 0 && clk) clk[16 * tile + (cphase++)] = (uint32_t)(wall_clock64() - c0); } while (0)
#else
#define TS_CLK() do {} while (0)
#endif
    uint32_t key[E], idx[E], rb[E];
    // span position p = e * 256 + tid: coalesced slot loads, one 16-byte gather per element, batches of up to 8 elements per lane in flight
    constexpr int BATCH = E < 8 ? E : 8;
    static_for<0, E / BATCH>([&](auto bc) {
        constexpr int e0 = BATCH * decltype(bc)::value;
        uint32_t slot[BATCH];
#pragma unroll
        for (int e = 0; e < BATCH; e++) {
            const uint32_t p = (uint32_t)((e0 + e) * 256 + tid);
            slot[e] = p < n ? slot_sorted[start + p] : 0u;
        }
        uint4 rc[BATCH];
#pragma unroll
        for (int e = 0; e < BATCH; e++) rc[e] = (uint32_t)((e0 + e) * 256 + tid) < n ? e_rec[slot[e]] : make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
        static_for<0, BATCH>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const uint32_t p = (uint32_t)((e0 + e) * 256 + tid);
            key[e0 + e] = rc[e].z;
            idx[e0 + e] = p;
            s_v[skew((int)p)] = rc[e].x;
            s_r[skew((int)p)] = rc[e].y & ~GS2M_ROWS_BIG;
            rb[e0 + e] = p < n && (rc[e].y & GS2M_ROWS_BIG) == 0u ? wave_rowbase[(rc[e].x & GS2M_GID_MASK) >> 6] : 0u;  // (a heavy instance's row is absolute)
        });
        asm volatile("" ::: "memory");  // the next batch's loads stay behind this batch's staging (registers: one batch in flight)
    });
    TS_CLK();  // 0: records staged (loads done)
    network_wg<E, LE>(key, idx, tid, s_x);
    TS_CLK();  // 1: sorted
#pragma unroll
    for (int e = 0; e < E; e++) s_r[skew(e * 256 + tid)] += rb[e];  // (this thread parked it: LDS operations of one wave execute in order)
    // ids and rows of the sorted elements: picked up by span position
    uint32_t sv[E], sr[E];
    auto pick_up = [&]() {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool real = (uint32_t)(tid * E + e) < n;
            sv[e] = real ? s_v[skew((int)idx[e])] : 0u;
            sr[e] = real ? s_r[skew((int)idx[e])] : 0u;
        }
    };
    auto park = [&]() {  // parked again under their sorted position
#pragma unroll
        for (int e = 0; e < E; e++) {
            s_v[skew(tid * E + e)] = sv[e];
            s_r[skew(tid * E + e)] = sr[e];
        }
    };
    gs2m_sync();  // (the row bases added above)
    pick_up();
    // equal depths anywhere in the sorted span (neighbours inside a lane, across lanes, across waves): tie_positions above
    bool tie = false;
#pragma unroll
    for (int e = 0; e + 1 < E; e++) tie |= (uint32_t)(tid * E + e + 1) < n && key[e] == key[e + 1];
    s_x[tid] = key[0];
    gs2m_sync();
    tie |= tid < 255 && (uint32_t)(tid * E + E) < n && key[E - 1] == s_x[tid + 1];
    if (gs2m_sync_or(tie) != 0) {  // (the barrier also closes the pick-up: the staging arrays are free)
#pragma unroll
        for (int e = 0; e < E; e++) {
            s_v[skew(tid * E + e)] = key[e];
            s_r[skew(tid * E + e)] = idx[e];
        }
        gs2m_sync();
        uint32_t pos[E];
        tie_positions<E>(key, idx, pos, tid, n, s_v, s_r);
        gs2m_sync();  // every look along a run is done: the arrays take the payload
#pragma unroll
        for (int e = 0; e < E; e++) {
            s_v[skew((int)pos[e])] = sv[e];
            s_r[skew((int)pos[e])] = sr[e];
        }
    } else {
        park();
    }
    TS_CLK();  // 2: ties settled
    gs2m_sync();
    TS_CLK();  // 3: payload in sorted order
    for (uint32_t k = (uint32_t)tid; k < n; k += 256) point_list[start + k] = s_v[skew((int)k)];
    // quadrant lists: wave q compacts quadrant q
    const int q = wave;
    uint2* out = qlist + (size_t)4 * start + (size_t)q * n;
    uint32_t* orow = qrow + (size_t)4 * start + (size_t)q * n;
    uint32_t run = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t base = 0; base < n; base += GS2M_WAVE) {
        const uint32_t k = base + (uint32_t)lane;
        uint32_t v = 0, r = 0;
        if (k < n) { v = s_v[skew((int)k)]; r = s_r[skew((int)k)]; }
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
    TS_CLK();  // 4: lists written (stores issued)
#ifdef GS2M_TS_CLOCK
    if (tid == 0 && clk) { clk[16 * tile + 14] = n; clk[16 * tile + 15] = (uint32_t)(c0 & 0xFFFFFFFFull); }
#endif
}

// More than 4096 entries (a tile under thousands of Gaussians: dense foliage in a multi-million-Gaussian scene).  The span is sorted in
// the tile's own (still unused) quadrant-list region in global memory (32 n bytes: keys K[n], span positions I[n], and I2[n] for the
// last step), but almost all of the network still runs in registers:
//   A. every chunk of 4096 entries is gathered and sorted by depth with the register network above (16 per lane), written out as a run;
//   B. merge level by level (runs of 4096 -> 8192 -> ...): the mirrored compare of a level and its strides of 4096 and more run over
//      global memory (one step for two runs, three for four, ...), the strides 2048 .. 1 again in registers, a chunk at a time;
//   C. equal depths: every entry looks along its run for its place (tie_positions' rule).
// Exercised by tests/test_tile_sort_gpu.py (spans of up to 20000 entries, with equal depths) and the dense-scene tests.
constexpr int BIG_CHUNK = 4096;
__device__ __forceinline__ void sort_tile_big(const int tile, const uint32_t start, const uint32_t n, const uint32_t* __restrict__ slot_sorted,
                                              const uint4* __restrict__ e_rec, const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list,
                                              uint32_t* __restrict__ row_tmp /* R words: sorted rows on their way to the lists */,
                                              uint2* __restrict__ qlist, uint32_t* __restrict__ qrow, uint32_t* __restrict__ qcount,
                                              uint32_t* s_x /* BIG_CHUNK + BIG_CHUNK / 32 words: the register network's exchange array */) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t* const K = reinterpret_cast<uint32_t*>(qlist + (size_t)4 * start);
    uint32_t* const I = K + n;
    auto barrier = [&]() {
        __threadfence_block();
        gs2m_sync();
    };
    constexpr int E = 16, LE = 4;
    // ---- A: sorted runs of 4096 ----
    for (uint32_t base = 0; base < n; base += BIG_CHUNK) {
        uint32_t key[E], idx[E];
        static_for<0, E / 8>([&](auto bc) {  // gathers in batches of 8 per lane
            constexpr int e0 = 8 * decltype(bc)::value;
            uint32_t slot[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t p = base + (uint32_t)((e0 + e) * 256 + tid);
                slot[e] = p < n ? slot_sorted[start + p] : 0u;
            }
            static_for<0, 8>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                const uint32_t p = base + (uint32_t)((e0 + e) * 256 + tid);
                key[e0 + e] = p < n ? e_rec[slot[e]].z : 0xFFFFFFFFu;  // (a real key is the bit pattern of a depth > 0.2: never all ones)
                idx[e0 + e] = p;
            });
        });
        network_wg<E, LE>(key, idx, tid, s_x);
#pragma unroll
        for (int e = 0; e < E; e++) {  // the padding sorts behind every real entry of the chunk
            const uint32_t i = base + (uint32_t)(tid * E + e);
            if (i < n) { K[i] = key[e]; I[i] = idx[e]; }
        }
    }
    barrier();
    // ---- B: merge levels ----
    auto cex = [&](uint32_t i, uint32_t p) {
        const uint32_t a = K[i], b = K[p];
        if (b < a) {
            const uint32_t xa = I[i], xb = I[p];
            K[i] = b; K[p] = a; I[i] = xb; I[p] = xa;
        }
    };
    for (int kb = 13; (1u << (kb - 1)) < n; kb++) {
        const uint32_t half = 1u << (kb - 1), mask = (1u << kb) - 1u;
        for (uint32_t q = tid;; q += 256) {  // mirrored compare inside blocks of 2^kb; pairs with the partner in the (virtual, +inf) padding are no-ops
            const uint32_t i = ((q >> (kb - 1)) << kb) | (q & (half - 1u));
            if (i >= n) break;
            const uint32_t p = i ^ mask;
            if (p < n) cex(i, p);
        }
        barrier();
        for (int jb = kb - 2; jb >= 12; jb--) {  // strides of 4096 and more
            const uint32_t j = 1u << jb;
            for (uint32_t q = tid;; q += 256) {
                const uint32_t i = ((q >> jb) << (jb + 1)) | (q & (j - 1u));
                if (i >= n) break;
                const uint32_t p = i | j;
                if (p < n) cex(i, p);
            }
            barrier();
        }
        for (uint32_t base = 0; base < n; base += BIG_CHUNK) {  // strides 2048 .. 1 inside every chunk, element tid * 16 + e of the chunk in register e
            uint32_t key[E], idx[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t i = base + (uint32_t)(tid * E + e);
                key[e] = i < n ? K[i] : 0xFFFFFFFFu;
                idx[e] = i < n ? I[i] : i;
            }
            cross_wave<E>(key, idx, tid, tid ^ 128, false, (wave & 2) == 0, s_x);
            cross_wave<E>(key, idx, tid, tid ^ 64, false, (wave & 1) == 0, s_x);
            xor_steps<E, LE, LE + 5>(key, idx, lane);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t i = base + (uint32_t)(tid * E + e);
                if (i < n) { K[i] = key[e]; I[i] = idx[e]; }
            }
        }
        barrier();
    }
    // ---- C: equal depths in span order ----
    uint32_t* const I2 = I + n;
    for (uint32_t p = tid; p < n; p += 256) {
        const uint32_t Kp = K[p], X = I[p];
        uint32_t a = p, cnt = 0;
        for (uint32_t j = p; j > 0u;) {
            j--;
            if (K[j] != Kp) break;
            a = j;
            cnt += I[j] < X ? 1u : 0u;
        }
        for (uint32_t j = p + 1u; j < n; j++) {
            if (K[j] != Kp) break;
            cnt += I[j] < X ? 1u : 0u;
        }
        I2[a + cnt] = X;
    }
    barrier();
    // sorted ids -> point_list (final), absolute rows -> row_tmp
    for (uint32_t p = tid; p < n; p += 256) {
        const uint4 rc = e_rec[slot_sorted[start + I2[p]]];
        point_list[start + p] = rc.x;
        row_tmp[start + p] = (rc.y & GS2M_ROWS_BIG) != 0u ? (rc.y & ~GS2M_ROWS_BIG) : rc.y + wave_rowbase[(rc.x & GS2M_GID_MASK) >> 6];
    }
    barrier();
    // quadrant lists: wave q compacts quadrant q (the scratch in the list region is dead behind the barrier above)
    const int q = wave;
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

// A workgroup per tile, in two launches: spans of up to 1024 entries (MAXE = 4: 13 KB of LDS and 80 registers, six workgroups per CU;
// this launch also writes ranges[] -- it is the frame's first when it runs at all) and the longer ones (MAXE = 16: 8 elements per lane up to 2048 entries, 16 up to 4096, 50 KB of LDS; launched on every
// frame).  A fixed grid walks the tiles: on a frame without such spans a launch costs what ~1000 workgroups cost to look at a few tile
// ranges each.
#ifndef GS2M_TS_WG_OCC
#define GS2M_TS_WG_OCC 6
#endif
template <int MAXE>
struct WgCfg {
    static constexpr int kMax = 256 * MAXE, kWords = kMax + kMax / 32;
    static constexpr int kLds = 3 * kWords;
};
template <int MAXE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MAXE >= 8 ? 3 : GS2M_TS_WG_OCC, 8)))
tile_sort_wg_kernel(const uint32_t* __restrict__ ranges_raw, uint2* __restrict__ ranges, const uint32_t* __restrict__ slot_sorted,
                    const uint4* __restrict__ e_rec, const uint32_t* __restrict__ wave_rowbase, uint32_t* __restrict__ point_list,
                    uint32_t* __restrict__ row_tmp, uint2* __restrict__ qlist, uint32_t* __restrict__ qrow, uint32_t* __restrict__ qcount,
                    int tiles_x, int tiles_y, uint32_t nblocks, uint32_t* __restrict__ counters, int take_mid) {
    constexpr int WORDS = WgCfg<MAXE>::kWords;
    __shared__ uint32_t s_all[WgCfg<MAXE>::kLds];
    if constexpr (MAXE >= 8) {
        // no span beyond 1024 entries in this frame (the kernel in front looked) -- and, when the spans of 513 .. 1024 entries are this
        // kernel's as well (take_mid: a frame whose average span is short has too few of them for a launch of their own), none of those
        if (counters[GS2M_CNT_SPAN_LONG] == 0u && !(take_mid && counters[GS2M_CNT_SPAN_MID] != 0u)) return;
    }
    uint32_t* const s_v = s_all, * const s_r = s_all + WORDS, * const s_x = s_all + 2 * WORDS;
    const int tid = threadIdx.x;
    for (uint32_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {  // (gridDim.x is a multiple of 8: a workgroup stays on its XCD's tiles)
        const int tile = tile_of_block((int)blk, tiles_x, tiles_y);
        if (tile < 0) continue;
        const uint2 raw = reinterpret_cast<const uint2*>(ranges_raw)[tile];
        const uint32_t start = raw.y != 0u ? ~raw.x : 0u, n = raw.y != 0u ? raw.y - start : 0u;
        if constexpr (MAXE < 8) {
            if (ranges != nullptr) {
                if (tid == 0) ranges[tile] = raw.y != 0u ? make_uint2(start, raw.y) : make_uint2(0u, 0u);
                if (n == 0u && tid < 4) qcount[tile * 4 + tid] = 0u;
            }
            if (n > 1024u && tid == 0) counters[GS2M_CNT_SPAN_LONG] = 1u;
            if (n == 0u || n > 1024u) continue;
            if (n <= 256u) sort_tile_wg<1, 0>(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, s_x, tid, row_tmp);
            else if (n <= 512u) sort_tile_wg<2, 1>(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, s_x, tid, row_tmp);
            else sort_tile_wg<4, 2>(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, s_x, tid, row_tmp);
        } else {
            if (n <= 512u || (n <= 1024u && !take_mid)) continue;
            if (n <= 1024u) sort_tile_wg<4, 2>(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, s_x, tid, row_tmp);
            else if (n <= 2048u) sort_tile_wg<8, 3>(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, s_x, tid, row_tmp);
            else if (n <= (uint32_t)WG_MAX) sort_tile_wg<16, 4>(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, qlist, qrow, qcount, s_v, s_r, s_x, tid, row_tmp);
            else sort_tile_big(tile, start, n, slot_sorted, e_rec, wave_rowbase, point_list, row_tmp, qlist, qrow, qcount, s_all);
        }
        gs2m_sync();  // the LDS arrays are the next tile's
    }
}

}  // namespace

static std::atomic<int> g_ts_policy{0};
void gs2m_set_tile_sort_policy_impl(int policy) { g_ts_policy = policy; }

void gs2m_launch_tile_sort(size_t tiles, int tiles_x, int tiles_y, size_t R /* instances of the frame; SIZE_MAX: unknown */, const BinningState& b, const ImageState& im,
                           const GeomState& g, hipStream_t s) {
    if (tiles == 0) return;
    const unsigned grid = tile_grid(tiles_x, tiles_y);
    // Who sorts which span.  A frame of many tiles (1080p: 8160) fills the chip with one wave per tile (8 elements per lane up to 512
    // entries, 16 up to 1024: 4096 to 8192 tiles resident, each with all its record gathers in flight -- the kernels are bound by those
    // gathers).  A frame of few tiles (a 777 x 581 training view: 1813) leaves such a kernel one or two waves per SIMD, each a serial
    // chain of ~40 / ~90 us: there every tile gets a workgroup (27 + 18 us instead of 40 + 99 + 55 on that view; on 8160 tiles of 400 to
    // 900 entries the workgroup kernels lose, 315 against 173 us: 1536 tiles resident instead of 4096).  policy: 0 = by tile count,
    // 1 = workgroups, 2 = waves, 3 = waves with the spans of 513 .. 1024 entries left to the workgroup kernel behind (what 0 and 2 do
    // by themselves in a frame whose AVERAGE span is short: see below).
    static const int env_policy = getenv("GS2M_TS_POLICY") ? atoi(getenv("GS2M_TS_POLICY")) : 0;  // (experiments)
    const int set_policy = g_ts_policy.load(std::memory_order_relaxed), policy = set_policy != 0 ? set_policy : env_policy;
    const bool waves = policy == 2 || policy == 3 || (policy == 0 && tiles >= 2560);
    // Spans of 513 .. 1024 entries under the wave policy: a kernel of their own, one wave per tile with 16 elements per lane -- launched
    // over ALL tiles, 5.4 us of dispatch when it finds nothing -- where they are common (an average span above 400: at C5's 660 every
    // tile is one); in a frame of short spans (the bench cloud: 332 on average, none above 512) the few that may exist go to the
    // workgroup kernel that takes the spans beyond 1024 anyway, which leaves at once when there is neither.
    const bool mid_own_kernel = policy != 3 && (R == SIZE_MAX || R > (size_t)400 * tiles);
    const unsigned wg_grid = grid < 2048u ? grid : 2048u;  // (both multiples of 8)
    if (waves) {
        tile_sort_wave_kernel<<<grid, 64, 0, s>>>(im.ranges_raw, im.ranges, b.slot_sorted, b.e_rec, g.wave_rowbase, b.point_list, b.qlist, b.qrow, im.qcount,
                                                  tiles_x, tiles_y, 512u, g.counters);
        if (mid_own_kernel)
            tile_sort_wave16_kernel<<<grid, 64, 0, s>>>(im.ranges_raw, b.slot_sorted, b.e_rec, g.wave_rowbase, b.point_list, b.qlist, b.qrow, im.qcount, tiles_x, tiles_y, g.counters);
    } else {
        tile_sort_wg_kernel<4><<<wg_grid, 256, 0, s>>>(im.ranges_raw, im.ranges, b.slot_sorted, b.e_rec, g.wave_rowbase, b.point_list, b.sort_valA, b.qlist, b.qrow,
                                                       im.qcount, tiles_x, tiles_y, grid, g.counters, 0);
    }
    // spans of more than 1024 entries: none on the bench scenes (the kernel in front says so: the workgroups leave at once)
    tile_sort_wg_kernel<16><<<wg_grid < 768u ? wg_grid : 768u /* three workgroups per CU: all resident at once */, 256, 0, s>>>(im.ranges_raw, nullptr, b.slot_sorted, b.e_rec, g.wave_rowbase, b.point_list, b.sort_valA,
                                                                             b.qlist, b.qrow, im.qcount, tiles_x, tiles_y, grid, g.counters, waves && !mid_own_kernel ? 1 : 0);
}
