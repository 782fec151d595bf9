// Stable LSD radix sort of (u32 key, u32 value) pairs for gfx950, onesweep style
// (Adinets & Merrill 2022): ONE histogram kernel reads the keys once and counts every digit of
// every pass; then one kernel per pass ranks and scatters a 4096-key tile per workgroup, obtaining
// the number of equal digits in all preceding tiles by decoupled look-back instead of a separate
// scan pass.  Used for (1) the depth order of the Gaussians (32-bit fp32 depth keys) and (2) the
// stable tile sort of the emitted instances (13 key bits at 1080p) -- the two places where the
// reference calls cub::DeviceRadixSort (cuda_rasterizer/rasterizer_impl.cu:291-296 sorts R 64-bit
// keys over 45 bits in one go).  Round 4: a sort's last pass can carry side jobs that replace kernels of their own -- the
// tile ranges (range_raw: identifyTileRanges) and block sums of a per-value term in final order (SideBuckets: the reference's
// InclusiveSum over tiles_touched); the histogram can come from the producer of the keys (ext_hist).
//
// MI355X specifics:
//  * 64-lane ranking: the lanes holding the same digit are found with BITS ballots (match-any);
//    rank = popcount of the lower peers; the lowest peer bumps the wave's digit counter in LDS.
//    Items of a lane are visited in index order, waves own contiguous key ranges, so the sort is
//    stable by construction.
//  * look-back across workgroups that may sit on different XCDs (non-coherent L2s): every status
//    word carries flag and count together in ONE naturally aligned 32-bit word, written and polled
//    with relaxed AGENT-scope atomics (sc1, served by the memory side), so no fence and no separate
//    payload are needed; tile ids are handed out by an atomic ticket, so a tile only ever waits for
//    tiles that were started before it (forward progress without co-residency assumptions).
#include "common.h"
#include <stdlib.h>
#include <atomic>

static std::atomic<int> g_sort_tickets{0};
extern "C" int gs2m_set_sort_tickets(int on) {
    g_sort_tickets = on ? 1 : 0;
    return GS2M_OK;
}

namespace {
#ifndef LB2_WIN
#define LB2_WIN 16  // groups looked at per round trip of the look-back
#endif

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 keys per workgroup
constexpr int RS_MAXPASS = 4;
// Passes of at most this many tiles take their tile ids from blockIdx (see launch_pass): 4 workgroups per compute
// unit of the CURRENT device (the kernel's LDS, 38 KB, and its <= 128 VGPRs allow 4) -- a
// partitioned device (CPX: 32 CUs) gets a proportionally smaller bound.
int resident_tiles() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int cus = 0;
        cached[dev] = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 ? 4 * cus : -1;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}
constexpr uint32_t FLAG_AGG = 1u << 30, FLAG_PFX = 2u << 30, VAL_MASK = (1u << 30) - 1;

struct SortPlan {
    int npass;
    int bits[RS_MAXPASS];
    int shift[RS_MAXPASS];
};

SortPlan make_plan(int total_bits) {
    SortPlan p;
    if (total_bits < 1) total_bits = 1;
    p.npass = total_bits <= 16 ? 2 : 4;  // even: the result lands back in the input buffers
    const int per = (total_bits + p.npass - 1) / p.npass;
    int s = 0;
    for (int i = 0; i < p.npass; i++) {
        int b = total_bits - s;
        if (b > per) b = per;
        if (b < 0) b = 0;
        p.bits[i] = b;
        p.shift[i] = s;
        s += b;
    }
    return p;
}

// ---- histogram of every digit of every pass, one read of the keys ----
// At most RS_HIST_WGS workgroups stride over the 4096-key tiles and keep their counts in LDS until the end, so the
// global histogram receives at most RS_HIST_WGS x (non-empty bins) atomic adds (one workgroup per tile sent 190 k adds
// to the 1024 words of the depth sort: same-address atomics serialise at the memory side).  The LDS counters are
// replicated RS_HIST_REP times (replica = lane & 15): depth keys of a scene share their top byte, and 64 lanes adding
// to ONE LDS word serialise 64-fold; with the replicas it is 4-fold at worst.  The side sum is kept per thread and
// reduced across the wave at the end (it used to be an LDS atomic on one word per key).
#ifndef RS_HIST_WGS_N
#define RS_HIST_WGS_N 512
#endif
constexpr int RS_HIST_WGS = RS_HIST_WGS_N;
constexpr int RS_HIST_REP = 16;
__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int npass,
                                                             int4 bits, int4 shift, uint32_t* __restrict__ ghist, int tiles,
                                                             int workers, SideSum sum) {
    __shared__ uint32_t s_h[RS_MAXPASS][256][RS_HIST_REP];
    __shared__ uint32_t s_sum;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_sum = 0;
    for (int i = tid; i < RS_MAXPASS * 256 * RS_HIST_REP; i += RS_THREADS) (&s_h[0][0][0])[i] = 0;
    gs2m_sync();
    const int b[4] = {bits.x, bits.y, bits.z, bits.w}, sh[4] = {shift.x, shift.y, shift.z, shift.w};
    const int rep = lane & (RS_HIST_REP - 1);
    uint32_t tsum = 0;
    for (int tile = blockIdx.x; tile < tiles; tile += workers) {
        const uint32_t base = (uint32_t)tile * RS_TILE;
        // all 16 keys (and side-sum terms) of the thread are requested before the first is used: one memory round trip per
        // tile instead of four (the kernel is a single short pass over the keys: latency is all it costs)
        uint32_t key[RS_ITEMS], tv[RS_ITEMS];
#pragma unroll
        for (int k = 0; k < RS_ITEMS; k++) {
            const uint32_t i = base + k * RS_THREADS + tid;
            key[k] = i < n ? keys[i] : 0u;
            tv[k] = (sum.tt && i < n) ? sum.tt[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < RS_ITEMS; k++) {
            const uint32_t i = base + k * RS_THREADS + tid;
            if (i < n) {
                for (int p = 0; p < npass; p++) atomicAdd(&s_h[p][(key[k] >> sh[p]) & ((1u << b[p]) - 1u)][rep], 1u);
                tsum += tv[k];
            }
        }
    }
    if (sum.tt) {
        tsum = wave_inclusive_scan_u32(tsum, lane);
        if (lane == 63 && tsum) atomicAdd(&s_sum, tsum);
    }
    gs2m_sync();
    if (sum.tt && tid == 0) {
        // the total of `tt` (num_rendered) leaves for the host as soon as the LAST workgroup has added its share: long
        // before the sort and the scan behind this kernel are done (api.hip).
        // ONE 64-bit returning atomic carries both the running total (low 44 bits: every partial sum is below 2^32 and at most
        // 512 workgroups add one) and the number of workgroups that have added theirs (bits 44 and up): the workgroup whose
        // add returns workers - 1 in the count field is the last one and holds the grand total -- no fence, no second
        // counter.  Published saturated, so a count beyond 2^32 cannot wrap past the caller's range check.  The words are
        // zeroed on the stream ahead of every call (api.hip: the preprocess kernel's zero jobs).
        unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(sum.acc);
        const unsigned long long old = __hip_atomic_fetch_add(acc64, (1ull << 44) | (unsigned long long)s_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(old >> 44) == (uint32_t)workers - 1u) {
            const unsigned long long total = (old & ((1ull << 44) - 1ull)) + (unsigned long long)s_sum;
            __hip_atomic_store(sum.landing, total > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (int p = 0; p < npass; p++) {
        const uint4* r4 = reinterpret_cast<const uint4*>(&s_h[p][tid][0]);
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < RS_HIST_REP / 4; q++) {
            const uint4 v = r4[q];
            c += v.x + v.y + v.z + v.w;
        }
        if (c) atomicAdd(&ghist[p * 256 + tid], c);
    }
}

// ---- one pass: rank + look-back + scatter ----
template <int BITS>
__global__ void __launch_bounds__(RS_THREADS) rs_onesweep_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, uint32_t n, int shift, const uint32_t* __restrict__ ghist, uint32_t* ticket,
    uint32_t* status /* [tiles][256] */, uint32_t* range_raw, int hist_copies /* ghist is the sum of this many copies, GS2M_HIST_COPY_WORDS apart */,
    SideBuckets sb) {
    constexpr int BINS = 1 << BITS;
    __shared__ uint32_t s_cnt[4][BINS];  // per-wave digit counters, later per-wave local bases
    __shared__ uint32_t s_gbase[256];    // global position of local index i with digit d: s_gbase[d] + i
    __shared__ uint32_t s_w[2][4];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_key[RS_TILE], s_val[RS_TILE];  // tile reordered by digit: coalesced runs on the way out
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_tile = ticket != nullptr ? atomicAdd(ticket, 1u) : blockIdx.x;
    for (int i = tid; i < 4 * BINS; i += RS_THREADS) (&s_cnt[0][0])[i] = 0;
    gs2m_sync();
    const uint32_t tile = s_tile;
    const uint32_t tile_base = tile * RS_TILE;
    const uint32_t tile_count = min((uint32_t)RS_TILE, n - tile_base);
    const uint32_t segbase = tile_base + wave * (64 * RS_ITEMS);

    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const uint32_t i = segbase + k * 64 + lane;
        key[k] = i < n ? keys_in[i] : 0xFFFFFFFFu;
        val[k] = i < n ? (vals_in ? vals_in[i] : i) : 0u;  // vals_in == nullptr: the value is the index
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const bool valid = segbase + k * 64 + lane < n;
        const uint32_t d = (key[k] >> shift) & (BINS - 1);
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t old = s_cnt[wave][d];  // every peer reads before the leader's write (in-order LDS)
        const unsigned long long below = peers & lt;
        rank[k] = old + (uint32_t)__popcll(below);
        if (valid && below == 0ull) s_cnt[wave][d] = old + (uint32_t)__popcll(peers);
    }
    gs2m_sync();

    // per digit: tile total, exclusive scans over the digits of (a) the tile totals (local layout) and
    // (b) the global histogram (digit bases), then the look-back for the tiles in front
    uint32_t tot = 0, c0 = 0, c1 = 0, c2 = 0, gh = 0;
    if (tid < BINS) {
        c0 = s_cnt[0][tid]; c1 = s_cnt[1][tid]; c2 = s_cnt[2][tid];
        tot = c0 + c1 + c2 + s_cnt[3][tid];
        for (int c = 0; c < hist_copies; c++) gh += ghist[c * GS2M_HIST_COPY_WORDS + tid];
    }
    const uint32_t inclA = wave_inclusive_scan_u32(tot, lane), inclB = wave_inclusive_scan_u32(gh, lane);
    if (lane == 63) {
        s_w[0][wave] = inclA;
        s_w[1][wave] = inclB;
    }
    gs2m_sync();
    uint32_t tile_excl = inclA - tot, digit_base = inclB - gh;
#pragma unroll
    for (int w = 0; w < 4; w++)
        if (w < wave) {
            tile_excl += s_w[0][w];
            digit_base += s_w[1][w];
        }
    // The tile's digit totals are published first (the tiles behind are waiting for them), then the tile is reordered by digit
    // in LDS -- which needs only the tile's own counts -- and only then do the digit threads look back: the keys, values and
    // ranks are out of the registers by then (128 instead of 162 VGPRs: four workgroups per compute unit instead of three).
    if (tid < BINS) {
        __hip_atomic_store(status + (size_t)tile * 256 + tid, FLAG_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_cnt[0][tid] = tile_excl;
        s_cnt[1][tid] = tile_excl + c0;
        s_cnt[2][tid] = tile_excl + c0 + c1;
        s_cnt[3][tid] = tile_excl + c0 + c1 + c2;
    }
    gs2m_sync();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        if (segbase + k * 64 + lane < n) {
            const uint32_t d = (key[k] >> shift) & (BINS - 1);
            const uint32_t lp = s_cnt[wave][d] + rank[k];
            s_key[lp] = key[k];
            s_val[lp] = val[k];
        }
    }
    if (tid < BINS) {
        // Where the tiles in front end, per digit.  Two levels, no chain through the tiles: every tile publishes its digit
        // totals; a tile's own offset is (the totals of the tiles in front of it in its GROUP of 32, read directly -- up to 31
        // independent loads) + (the totals of the groups in front).  A group's total is published by its LAST tile, which has
        // it once it has read its 31 predecessors, followed by the group's inclusive prefix; every tile adds the group totals
        // back to the nearest published prefix, 16 groups per round trip.  Rounds 1-4 walked back over the tiles themselves,
        // 4 per round trip (with every tile of a pass resident and publishing at the same moment the prefixes spread as
        // ~2 j^2 tiles after j round trips): the instance sort's passes took 31 us, now 28; the depth sort's 21, now 20;
        // a pass without any look-back (wrong positions, timing only) takes 18 / 13.
        const uint32_t grp = tile >> 5, r = tile & 31u;
        uint32_t* const lvl2 = status + (size_t)gridDim.x * 256;  // [groups][256] behind the [tiles][256] totals
        const bool leader = r == 31u;
        uint32_t insum = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t w[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t back = (uint32_t)(16 * h + k) + 1u;
                w[k] = back <= r ? __hip_atomic_load(status + (size_t)(tile - back) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FLAG_AGG;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t back = (uint32_t)(16 * h + k) + 1u;
                while ((w[k] & ~VAL_MASK) == 0u) {  // not published yet
                    __builtin_amdgcn_s_sleep(1);
                    w[k] = __hip_atomic_load(status + (size_t)(tile - back) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                insum += w[k] & VAL_MASK;
            }
        }
        uint32_t* const my2 = lvl2 + (size_t)grp * 256 + tid;
        const uint32_t gagg = insum + tot;
        if (leader) __hip_atomic_store(my2, (grp == 0 ? FLAG_PFX : FLAG_AGG) | gagg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the groups in front, LB2_WIN per round trip (independent loads): totals are added up to the nearest group whose
        // last tile has already published its prefix -- with the few dozen groups of a pass usually ONE round trip
        uint32_t excl2 = 0;
        {
            int p = (int)grp - 1;
            bool found = p < 0;
            while (!found) {
                uint32_t w[LB2_WIN];
#pragma unroll
                for (int k = 0; k < LB2_WIN; k++)
                    w[k] = p - k >= 0 ? __hip_atomic_load(lvl2 + (size_t)(p - k) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : FLAG_PFX;  // in front of group 0: prefix 0
                bool stalled = false;
#pragma unroll
                for (int k = 0; k < LB2_WIN; k++) {
                    const uint32_t f = w[k] & ~VAL_MASK;
                    if (!found && !stalled) {
                        if (f == 0u) {
                            stalled = true;  // not published yet: poll again from here
                        } else {
                            excl2 += w[k] & VAL_MASK;
                            p--;
                            found = f == FLAG_PFX;
                        }
                    }
                }
                if (stalled) __builtin_amdgcn_s_sleep(1);
            }
        }
        if (leader && grp > 0) __hip_atomic_store(my2, FLAG_PFX | (excl2 + gagg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t excl = excl2 + insum;
        s_gbase[tid] = digit_base + excl - tile_excl;
    }
    gs2m_sync();
    // side job, second level: the sums over super-blocks of 65536 positions are collected per workgroup in LDS first (the
    // per-wave counters are dead from here on: their 1024 words are reused) -- a workgroup's elements land in a handful of
    // super-blocks, and tens of thousands of atomics on so few words ran at ~80 ns apiece: 0.19 ms
    uint32_t* const s_sup = &s_cnt[0][0];
    constexpr uint32_t SUP_LDS = 4 * BINS < 1024 ? 4 * BINS : 1024;
    if (sb.tt != nullptr) {
        for (int q = tid; q < (int)SUP_LDS; q += RS_THREADS) s_sup[q] = 0u;
        gs2m_sync();
    }
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const uint32_t i = k * RS_THREADS + tid;
        uint32_t bpos = 0u, bval = 0u;
        if (i < tile_count) {
            const uint32_t kk = s_key[i];
            const uint32_t pos = s_gbase[(kk >> shift) & (BINS - 1)] + i;
            keys_out[pos] = kk;
            vals_out[pos] = s_val[i];
            if (range_raw != nullptr) {
                // the last pass of the sort: equal full keys are contiguous in the reordered tile (stable passes), so the
                // elements at the ends of a run know where the run begins / ends in the output; runs of one key in several
                // workgroup tiles combine through the atomics
                if (i == 0 || s_key[i - 1] != kk) atomicMax(&range_raw[2 * kk], ~pos);
                if (i + 1 == tile_count || s_key[i + 1] != kk) atomicMax(&range_raw[2 * kk + 1], pos + 1u);
            }
            bpos = pos;
            bval = s_val[i];
        }
        if (sb.tt != nullptr) {
            // Side job of the LAST pass of the depth sort: sums of tt[value] over blocks of 256 FINAL positions, which is
            // all the emit kernel needs from the reference's InclusiveSum (rasterizer_impl.cu:265-266) beyond its own 256
            // counts -- the separate scan kernel of rounds 1-3 (20 us + a launch boundary) is gone.  The wave's 64
            // elements are consecutive in the reordered tile, so their final positions increase: equal blocks are
            // contiguous lanes, summed by a segmented scan, one atomic per (wave step, block).
            uint32_t t = 0u, bkt = 0xFFFFFFFFu;
            if (i < tile_count) {
                t = sb.tt[bval];  // (gathering these at the top of the kernel and carrying them through LDS was slower: +16 KB of LDS in every pass)
                bkt = bpos >> 8;
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t tv = __shfl_up(t, d, 64), bv = __shfl_up(bkt, d, 64);
                if (lane >= d && bv == bkt) t += tv;
            }
            const uint32_t bn = __shfl_down(bkt, 1, 64);
            if (bkt != 0xFFFFFFFFu && (lane == 63 || bn != bkt) && t != 0u) {
                atomicAdd(&sb.buckets[bkt], t);
                // second level: sums over 256 blocks, so that a reader adds at most 256 + 256 words
                if ((bkt >> 8) < SUP_LDS) atomicAdd(&s_sup[bkt >> 8], t);
                else atomicAdd(&sb.supers[bkt >> 8], t);
            }
        }
    }
    if (sb.tt != nullptr) {
        gs2m_sync();
        for (int q = tid; q < (int)SUP_LDS; q += RS_THREADS)
            if (s_sup[q] != 0u) atomicAdd(&sb.supers[q], s_sup[q]);
    }
}

template <int BITS>
void launch_pass(const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, uint32_t n, int shift,
                 const uint32_t* ghist, uint32_t* ticket, uint32_t* status, int tiles, hipStream_t s, uint32_t* range_raw, int hist_copies, SideBuckets sb) {
    // Tile ids: blockIdx when every workgroup of the pass fits on the device at once with room to spare (then no tile
    // waits for one that cannot start, as long as the device is not shared with another resident kernel -- the call is
    // stream-ordered, and a look-back that does stall is still released by the dispatch of the earlier blocks, which
    // the hardware starts in id order); beyond that an atomic ticket hands them out in start order.  The ticket
    // serialises the starts on one address: 0.088 -> 0.074 ms for the 661-tile instance sort without it.
    // GS2M_SORT_TICKETS=1 or gs2m_set_sort_tickets(1) forces tickets: a device shared with other resident kernels -- RCCL's, in
    // data-parallel runs (gs2m_dp switches them on when it creates a reducer) -- or CU-masked.
    static const bool env_tickets = getenv("GS2M_SORT_TICKETS") && atoi(getenv("GS2M_SORT_TICKETS")) != 0;
    const bool force_tickets = env_tickets || g_sort_tickets.load(std::memory_order_relaxed) != 0;
    rs_onesweep_kernel<BITS><<<tiles, RS_THREADS, 0, s>>>(ki, vi, ko, vo, n, shift, ghist,
                                                          (!force_tickets && tiles <= resident_tiles()) ? nullptr : ticket, status, range_raw, hist_copies, sb);
}

}  // namespace

void gs2m_radix_plan(int total_bits, int* npass, int bits[4], int shift[4]) {
    const SortPlan p = make_plan(total_bits);
    *npass = p.npass;
    for (int i = 0; i < RS_MAXPASS; i++) { bits[i] = i < p.npass ? p.bits[i] : 0; shift[i] = i < p.npass ? p.shift[i] : 0; }
}

// temp layout: [ghist 4x256][tickets 4 (padded to 256 B)][status npass x (tiles + groups of 32 tiles) x 256]
static size_t status_rows(size_t tiles) { return tiles + (tiles + 31) / 32; }
size_t gs2m_radix_temp_bytes(size_t n, int total_bits) {
    const SortPlan p = make_plan(total_bits);
    const size_t tiles = (n + RS_TILE - 1) / RS_TILE;
    return gs2m_align_up(RS_MAXPASS * 256 * 4) + GS2M_ALIGN + (size_t)p.npass * status_rows(tiles) * 256 * 4 + 2 * GS2M_ALIGN;
}

// Sorts n pairs by key bits [0, total_bits), stable.  The input (kin, vin) is only read (vin may be
// nullptr: values are then the indices 0..n-1); passes alternate between (kA, vA) and (kB, vB) and the
// number of passes is even, so the result is in (kB, vB).
// The part of `temp` a sort of n keys needs zeroed beforehand (histograms, tickets, look-back status).  A caller that
// zeroes it itself -- e.g. inside the kernel that produces the keys -- passes prezeroed = true and saves a launch.
void gs2m_radix_zero_region(void* temp, size_t n, int total_bits, uint32_t** ptr, size_t* words) {
    const SortPlan p = make_plan(total_bits);
    const size_t tiles = (n + RS_TILE - 1) / RS_TILE;
    *ptr = (uint32_t*)gs2m_align_up((size_t)(uintptr_t)temp);
    *words = (gs2m_align_up(RS_MAXPASS * 256 * 4) + GS2M_ALIGN + (size_t)p.npass * status_rows(tiles) * 256 * 4) / 4;
}

hipError_t gs2m_radix_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* kin, const uint32_t* vin, uint32_t* kA,
                                 uint32_t* vA, uint32_t* kB, uint32_t* vB, size_t n, int total_bits, bool prezeroed, hipStream_t s,
                                 SideSum sum, uint32_t* range_raw, const uint32_t* ext_hist, SideBuckets sb) {
    if (n == 0) return hipSuccess;
    const SortPlan p = make_plan(total_bits);
    const int tiles = (int)((n + RS_TILE - 1) / RS_TILE);
    if (temp_bytes < gs2m_radix_temp_bytes(n, total_bits)) return hipErrorInvalidValue;
    char* base = (char*)gs2m_align_up((size_t)(uintptr_t)temp);
    uint32_t* ghist = (uint32_t*)base;
    uint32_t* tickets = (uint32_t*)(base + gs2m_align_up(RS_MAXPASS * 256 * 4));
    uint32_t* status = (uint32_t*)(base + gs2m_align_up(RS_MAXPASS * 256 * 4) + GS2M_ALIGN);
    const size_t zero_bytes = gs2m_align_up(RS_MAXPASS * 256 * 4) + GS2M_ALIGN + (size_t)p.npass * status_rows((size_t)tiles) * 256 * 4;
    hipError_t e = prezeroed ? hipSuccess : gs2m_zero_async(base, zero_bytes, s);
    if (e != hipSuccess) return e;
    const int workers = tiles < RS_HIST_WGS ? tiles : RS_HIST_WGS;
    // ext_hist: the producer of the keys has already counted every digit of every pass (binning.hip: emit_kernel, into
    // GS2M_HIST_COPIES copies to spread its atomics): no histogram kernel, no second read of the keys
    if (ext_hist == nullptr)
        rs_hist_kernel<<<workers, RS_THREADS, 0, s>>>(kin, (uint32_t)n, p.npass, make_int4(p.bits[0], p.bits[1], p.bits[2], p.bits[3]),
                                                      make_int4(p.shift[0], p.shift[1], p.shift[2], p.shift[3]), ghist, tiles, workers, sum);
    const uint32_t *ki = kin, *vi = vin;
    for (int i = 0; i < p.npass; i++) {
        uint32_t* ko = (i & 1) ? kB : kA;
        uint32_t* vo = (i & 1) ? vB : vA;
        uint32_t* st = status + (size_t)i * status_rows((size_t)tiles) * 256;
        const uint32_t* gh = (ext_hist ? ext_hist : ghist) + i * 256;
        const int hc = ext_hist ? GS2M_HIST_COPIES : 1;
        switch (p.bits[i]) {
#define RS_CASE(B) case B: launch_pass<B>(ki, vi, ko, vo, (uint32_t)n, p.shift[i], gh, tickets + i, st, tiles, s, i == p.npass - 1 ? range_raw : nullptr, hc, i == p.npass - 1 ? sb : SideBuckets{nullptr, nullptr, nullptr}); break;
            RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
#undef RS_CASE
            default: launch_pass<1>(ki, vi, ko, vo, (uint32_t)n, 31, gh, tickets + i, st, tiles, s, i == p.npass - 1 ? range_raw : nullptr, hc, i == p.npass - 1 ? sb : SideBuckets{nullptr, nullptr, nullptr}); break;  // 0 bits: stable copy
        }
        ki = ko;
        vi = vo;
    }
    return hipGetLastError();
}

// ---- scratch the preprocess kernel zeroes for the binning front end: [block sums of tiles_touched in depth order, one per
// 256 Gaussians (filled by the depth sort's last pass, read by emit_kernel)][the tile sort's digit histograms, GS2M_HIST_COPIES
// copies (filled by emit_kernel)] ----
// [block sums: (n + 255) / 256 + 64 words][super-block sums (256 blocks each): (n + 65535) / 65536 + 64 words][histograms]
static size_t front_blocks(size_t n) { return (n + 255) / 256 + 64; }
static size_t front_supers(size_t n) { return (n + 65535) / 65536 + 64; }
size_t gs2m_front_temp_bytes(size_t n) { return gs2m_align_up((front_blocks(n) + front_supers(n) + GS2M_HIST_COPIES * GS2M_HIST_COPY_WORDS) * 4) + 2 * GS2M_ALIGN; }
uint32_t* gs2m_block_sums_ptr(void* front_temp) { return (uint32_t*)gs2m_align_up((size_t)(uintptr_t)front_temp); }
uint32_t* gs2m_super_sums_ptr(void* front_temp, size_t n) { return gs2m_block_sums_ptr(front_temp) + front_blocks(n); }
uint32_t* gs2m_tile_hist_ptr(void* front_temp, size_t n) { return gs2m_super_sums_ptr(front_temp, n) + front_supers(n); }
void gs2m_front_zero_region(void* front_temp, size_t n, uint32_t** ptr, size_t* words) {
    *ptr = gs2m_block_sums_ptr(front_temp);
    *words = front_blocks(n) + front_supers(n) + GS2M_HIST_COPIES * GS2M_HIST_COPY_WORDS;
}
