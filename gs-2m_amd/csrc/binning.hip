// Tile binning for gfx950 (round 5: no sort of the Gaussians; index-order emission).
//
// The reference (cuda_rasterizer/rasterizer_impl.cu:63-103, 265-305) scans tiles_touched, expands every visible Gaussian into
// (tile << 32 | depth) u64 keys and runs one 45-bit radix sort over all R instances, then finds the tile ranges in the sorted
// keys.  Rounds 1-4 here sorted the P Gaussians by depth first (a histogram kernel + 4 radix passes whose cost is look-back
// latency), emitted instances in depth order and sorted them stably by tile.  Round 5 drops the depth sort:
//   blockscan_kernel  one workgroup: num_rendered = the sum of the per-block instance counts the preprocess kernel left,
//                 published to the host at once; their exclusive prefix (the reference's InclusiveSum at block granularity);
//   emit_kernel   load-balanced expansion in INDEX order (a wave owns 64 consecutive Gaussians and writes 64 consecutive
//                 slots per step): tile id per slot, exact ellipse-vs-8x8 test per instance -> 4-bit quadrant mask above the id,
//                 gradient rows numbered densely per wave; counts the tile sort's digits;
//   emit_heavy_kernel  the instances of HEAVY Gaussians (common.h: GS2M_HEAVY_TILES), a wave per unit of 64 instances: balanced
//                 whatever the index order of the big splats;
//   rowscan_kernel  exclusive prefix of the waves' row counts: rows are dense over the whole view (the backward's scratch is
//                 sized by their number, published to the host);
//   the stable 13-bit radix sort of (tile, slot) pairs (radix_sort.hip, two passes; its last pass records the tile ranges:
//                 identifyTileRanges, rasterizer_impl.cu:108-129) leaves every tile's span in index order;
//   tile_sort.hip orders each span by (depth, id) ON CHIP -- ties fall back on the span's own order, i.e. the Gaussian id:
//                 exactly the reference's order within a tile -- and splits it into the quadrant lists.
// A counting sort by tile with global atomics (count -> scan -> fill) was built first and measured: profiles/r05_frontend_ab.md.
#include "common.h"

GeomState gs2m_carve_geom(char* base, size_t P) {
    GeomState g;
    size_t off = base ? (gs2m_align_up((size_t)(uintptr_t)base) - (size_t)(uintptr_t)base) : 0;
    auto take = [&](size_t bytes) {  // base == nullptr: the returned "pointers" are byte offsets
        char* p = (char*)((uintptr_t)base + off);
        off = gs2m_align_up(off + bytes);
        return p;
    };
    const size_t nb = (P + 255) / 256 + 1, nw = (P + 63) / 64 + 1;
    g.rec = (float4*)take(P * REC_Q * sizeof(float4));
    g.tiles_touched = (uint32_t*)take(P * 4);
    g.rect = (uint2*)take(P * sizeof(uint2));
    g.depth_key = (uint32_t*)take(P * 4);
    g.clamped = (uint8_t*)take(P);
    g.sh_dir = (float*)take(P * 9 * sizeof(float));
    g.counters = (uint32_t*)take(64 * 4);
    g.gauss_rows = (uint32_t*)take(P * 4);
    g.block_tt = (uint32_t*)take(nb * 4);
    g.block_pref = (uint32_t*)take(nb * 4);
    g.block_hu = (uint32_t*)take(nb * 4);
    g.block_hupref = (uint32_t*)take(nb * 4);
    g.wave_rows = (uint32_t*)take(nw * 4);
    g.wave_rowbase = (uint32_t*)take(nw * 4);
    g.tile_hist = (uint32_t*)take(GS2M_HIST_COPIES * GS2M_HIST_COPY_WORDS * 4);
    g.total_bytes = off + GS2M_ALIGN;
    return g;
}

BinningState gs2m_carve_binning(char* base, size_t R, size_t temp_bytes, size_t heavy_units) {
    BinningState b;
    size_t off = base ? (gs2m_align_up((size_t)(uintptr_t)base) - (size_t)(uintptr_t)base) : 0;
    auto take = [&](size_t bytes) {
        char* p = (char*)((uintptr_t)base + off);
        off = gs2m_align_up(off + bytes);
        return p;
    };
    b.keys_unsorted = (uint32_t*)take(R * 4);
    b.e_rec = (uint4*)take(R * sizeof(uint4));
    b.sort_keyA = (uint32_t*)take(R * 4);
    b.sort_valA = (uint32_t*)take(R * 4);
    b.tile_keys = (uint32_t*)take(R * 4);
    b.slot_sorted = (uint32_t*)take(R * 4);
    b.point_list = (uint32_t*)take(R * 4);
    b.qlist = (uint2*)take(R * 4 * sizeof(uint2));
    b.qrow = (uint32_t*)take(R * 4 * 4);
    b.temp = take(temp_bytes);
    b.temp_bytes = temp_bytes;
    b.hrec = (HeavyUnit*)take(heavy_units * sizeof(HeavyUnit));
    b.total_bytes = off + GS2M_ALIGN;
    return b;
}

size_t gs2m_binning_temp_bytes(size_t R, int tile_bits) { return gs2m_align_up(gs2m_radix_temp_bytes(R, tile_bits)) + GS2M_ALIGN; }

ImageState gs2m_carve_image(char* base, size_t N, size_t tiles) {
    ImageState im;
    size_t off = base ? (gs2m_align_up((size_t)(uintptr_t)base) - (size_t)(uintptr_t)base) : 0;
    auto take = [&](size_t bytes) {
        char* p = (char*)((uintptr_t)base + off);
        off = gs2m_align_up(off + bytes);
        return p;
    };
    im.final_T = (float*)take(N * 4);
    im.n_contrib = (uint32_t*)take(N * 4);
    im.ranges = (uint2*)take(tiles * sizeof(uint2));
    im.ranges_raw = (uint32_t*)take(tiles * 2 * sizeof(uint32_t));
    im.qcount = (uint32_t*)take(tiles * 4 * sizeof(uint32_t));
    im.qlast = (uint32_t*)take(tiles * 4 * sizeof(uint32_t));
    im.total_bytes = off + GS2M_ALIGN;
    return im;
}

namespace {

__device__ __forceinline__ void publish(uint32_t* landing, int word, uint32_t v) {
    // one aligned system-scope 32-bit store into the mapped pinned block the host polls (a 4-byte hipMemcpyAsync may be
    // carried out byte by byte: torn counts were seen).  RELAXED: the host needs this word and nothing else of the kernel's
    // output -- a release at system scope writes the whole L2 back first (10 us behind a kernel that left tens of MB dirty)
    __hip_atomic_store(landing + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum over the workgroup (any size that is a multiple of 64, up to 1024 threads); valid in every thread.
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* s_part /* [16] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    gs2m_sync();  // s_part may still be read from a previous call
    if (lane == 0) s_part[wave] = v;
    gs2m_sync();
    unsigned long long t = 0;
    for (int w = 0; w < nw; w++) t += s_part[w];
    return t;
}

// ---- prefix sums by one workgroup ---------------------------------------------------------------------------------------
// Exclusive prefix sums of a few thousand words (3907 block counts / 15625 wave row counts at 1M Gaussians) WITHOUT a chain between
// workgroups and without a serial pass: workgroup b owns elements [1024 b, 1024 b + 1024) and adds up everything in front of them
// itself (coalesced, all loads requested together: one memory round trip; the whole array is tens of KB, read from L2 by every
// workgroup).  A single workgroup walking the array in rounds took 16 us for 15625 words: two dependent round trips behind a
// kernel that has just left tens of MB dirty in the L2s.  Beyond 64 workgroups (65536 elements: 16 M Gaussians for the block
// counts, 4 M for the wave rows) the part in front is read in rounds.
template <typename F>
__device__ __forceinline__ unsigned long long scan_1024_per_block(const uint32_t* __restrict__ in, size_t n, uint32_t* s_w /* [16] */,
                                                                  unsigned long long* s_part /* [16] */, F emit) {
    const size_t first = (size_t)blockIdx.x * 1024;
    unsigned long long before = 0;
    for (size_t base = 0; base < first; base += 16 * 1024) {  // the elements in front, 16 per thread and round
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const size_t i = base + (size_t)k * 1024 + threadIdx.x;
            v[k] = i < first ? in[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) before += v[k];
    }
    const size_t i = first + threadIdx.x;
    const uint32_t mine = i < n ? in[i] : 0u;
    before = block_sum_u64(before, s_part);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_scan_u32(mine, lane);
    if (lane == 63) s_w[wave] = incl;
    gs2m_sync();
    uint32_t run = incl - mine, total = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) run += s_w[w];
        total += s_w[w];
    }
    if (i < n) emit(i, (uint32_t)before + run, mine);
    return before + total;  // everything up to the end of this workgroup's elements
}

// the same for two arrays at once (block instance counts and block heavy-unit counts): one pass, one set of barriers
template <typename F>
__device__ __forceinline__ void scan2_1024_per_block(const uint32_t* __restrict__ in0, const uint32_t* __restrict__ in1, size_t n, uint32_t* s_w /* [32] */,
                                                     unsigned long long* s_part /* [32] */, F emit, unsigned long long& total0, unsigned long long& total1) {
    const size_t first = (size_t)blockIdx.x * 1024;
    unsigned long long b0 = 0, b1 = 0;
    for (size_t base = 0; base < first; base += 8 * 1024) {  // the elements in front, 8 of either array per thread and round
        uint32_t v[8], w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t i = base + (size_t)k * 1024 + threadIdx.x;
            v[k] = i < first ? in0[i] : 0u;
            w[k] = i < first ? in1[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { b0 += v[k]; b1 += w[k]; }
    }
    const size_t i = first + threadIdx.x;
    const uint32_t m0 = i < n ? in0[i] : 0u, m1 = i < n ? in1[i] : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { b0 += __shfl_xor(b0, d, 64); b1 += __shfl_xor(b1, d, 64); }
    const uint32_t i0 = wave_inclusive_scan_u32(m0, lane), i1 = wave_inclusive_scan_u32(m1, lane);
    if (lane == 0) { s_part[wave] = b0; s_part[16 + wave] = b1; }
    if (lane == 63) { s_w[wave] = i0; s_w[16 + wave] = i1; }
    gs2m_sync();
    unsigned long long f0 = 0, f1 = 0;
    uint32_t r0 = i0 - m0, r1 = i1 - m1, t0 = 0, t1 = 0;
    for (int w = 0; w < 16; w++) {
        f0 += s_part[w]; f1 += s_part[16 + w];
        if (w < wave) { r0 += s_w[w]; r1 += s_w[16 + w]; }
        t0 += s_w[w]; t1 += s_w[16 + w];
    }
    if (i < n) emit(i, (uint32_t)f0 + r0, (uint32_t)f1 + r1);
    total0 = f0 + t0;
    total1 = f1 + t1;
}

// num_rendered and the block prefixes.  The workgroup that owns the last elements has the grand total: it tells the host.
__global__ void __launch_bounds__(1024) blockscan_kernel(const uint32_t* __restrict__ block_tt, const uint32_t* __restrict__ block_hu, size_t nblocks,
                                                         uint32_t* __restrict__ block_pref, uint32_t* __restrict__ block_hupref,
                                                         uint32_t* __restrict__ counters, uint32_t* landing) {
    __shared__ uint32_t s_w[32];
    __shared__ unsigned long long s_part[32];
    unsigned long long t = 0, u = 0;
    scan2_1024_per_block(block_tt, block_hu, nblocks, s_w, s_part, [&](size_t b, uint32_t e0, uint32_t e1) { block_pref[b] = e0; block_hupref[b] = e1; }, t, u);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        // saturated: a count beyond 2^32 cannot wrap past the caller's range check.  num_rendered and the heavy units leave in ONE
        // 8-byte store: the host that sees the first has the second
        const uint32_t r32 = t > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)t, u32 = u > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)u;
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(landing + GS2M_LAND_R), (unsigned long long)r32 | ((unsigned long long)u32 << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        counters[1] = r32;
        counters[GS2M_CNT_HUNITS] = u32;
        counters[GS2M_CNT_SPAN_MID] = 0u;
        counters[GS2M_CNT_SPAN_LONG] = 0u;
    }
}

// exclusive prefix of the waves' gradient-row counts -> first row of every wave; the total goes to the host
__global__ void __launch_bounds__(1024) rowscan_kernel(const uint32_t* __restrict__ wave_rows, size_t nwaves, uint32_t* __restrict__ wave_rowbase,
                                                       uint32_t* __restrict__ counters, uint32_t* landing) {
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_part[16];
    // the heavy units' rows come first (4 x 64 per unit), the waves' dense rows behind them
    const uint32_t heavy_rows = 4u * GS2M_UNIT * counters[GS2M_CNT_HUNITS];
    const unsigned long long total = heavy_rows + scan_1024_per_block(wave_rows, nwaves, s_w, s_part, [&](size_t w, uint32_t excl, uint32_t) { wave_rowbase[w] = heavy_rows + excl; });
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        counters[GS2M_CNT_ROWS] = (uint32_t)total;
        publish(landing, GS2M_LAND_ROWS, (uint32_t)total + 1u);
    }
}

// ---- emit: instances in index order ---------------------------------------------------------------------------------------
// Load-balanced expansion (replaces duplicateWithKeys, rasterizer_impl.cu:63-103).  One wave owns 64 consecutive Gaussians
// whose instances occupy one contiguous slot range (first slot of the workgroup = block_pref[] + a scan of its own 256 counts);
// each step the wave writes 64 consecutive slots, each lane locating its source Gaussian by binary search in the wave's
// prefix sums.  Every instance is tested against the four 8x8 quadrants of its tile with the exact ellipse-vs-rectangle test
// (common.h) -- here the Gaussian's geometry is loaded once per Gaussian -- and the 4-bit hit mask travels above the Gaussian
// id.  The backward writes one gradient row per set bit; the rows of a wave's 64 Gaussians are numbered densely in emission
// order (Gaussian by Gaussian, instance by instance, quadrant by quadrant), relative to the wave's first row, which
// rowscan_kernel supplies afterwards (wave_rowbase) and tile_sort.hip adds: all rows of a Gaussian are one dense run and the
// per-Gaussian backward streams them (gaussian_bwd.hip).  The digits of the tile ids are counted here for the tile sort
// (radix_sort.hip: ext_hist), and the kernel zeroes that sort's scratch and the tile ranges on the side.
// HEAVY Gaussians (common.h: at least GS2M_HEAVY_TILES instances) are left out of the wave's loop: the wave only writes where their
// units start (gauss_rows) and whose the units are (HeavyUnit), emit_heavy_kernel below expands them.

// tile and quadrant-hit mask of instance t of a Gaussian: {tile id, mask}
__device__ __forceinline__ uint2 expand_instance(uint32_t rmin, uint32_t rwid, uint32_t t, const float4 a /* x, y, A, B */, const float2 ct /* C, t2 */,
                                                 int W, int H, int tiles_x) {
    const uint32_t ry = t / rwid, rx = t - ry * rwid;
    const uint32_t tx = (rmin & 0xFFFFu) + rx, ty = (rmin >> 16) + ry;
    const int px0 = (int)tx * GS2M_TILE, py0 = (int)ty * GS2M_TILE;
    uint32_t mask = gs2m_reaches_quads(a.x, a.y, a.z, a.w, ct.x, ct.y, (float)px0, (float)py0);
    // quadrants outside the image have no pixels: no list entry, no gradient row
    if (px0 + 8 >= W) mask &= 0x5u;
    if (py0 + 8 >= H) mask &= 0x3u;
    return make_uint2(ty * (uint32_t)tiles_x + tx, mask);
}

__global__ void __launch_bounds__(256) emit_kernel(int P, int W, int H, int tiles_x, const uint2* __restrict__ rect,
                                                   const uint32_t* __restrict__ block_pref, const uint32_t* __restrict__ block_hupref,
                                                   const float4* __restrict__ rec,
                                                   const uint32_t* __restrict__ depth_key, uint32_t* __restrict__ keys_out, uint4* __restrict__ e_rec,
                                                   HeavyUnit* __restrict__ hrec,
                                                   uint32_t* __restrict__ gauss_rows, uint32_t* __restrict__ wave_rows,
                                                   uint32_t* __restrict__ counters, uint32_t* __restrict__ tile_hist, int npass, int4 hbits,
                                                   int4 hshift, uint32_t crowded, ZeroJobs zero) {
    __shared__ uint32_t s_th[4][256];  // digit counts of this workgroup's keys, for the tile sort
    __shared__ uint32_t s_pref[4][GS2M_WAVE];
    __shared__ uint32_t s_rmin[4][GS2M_WAVE];
    __shared__ uint32_t s_rw[4][GS2M_WAVE];
    __shared__ uint32_t s_off[4][GS2M_WAVE];    // first emission slot of the Gaussian
    __shared__ uint32_t s_depth[4][GS2M_WAVE];  // its depth key
    __shared__ float4 s_geo[4][GS2M_WAVE];      // x, y, A, B
    __shared__ float2 s_ct[4][GS2M_WAVE];       // C, t2
    __shared__ uint32_t s_rc[4][GS2M_WAVE];     // gradient rows per Gaussian
    __shared__ uint32_t s_wtot[4], s_whu[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    gs2m_zero_jobs(zero, (size_t)i, (size_t)gridDim.x * 256);  // tile-sort scratch and the tile ranges
#pragma unroll
    for (int p = 0; p < 4; p++) s_th[p][threadIdx.x] = 0u;
    const int hb[4] = {hbits.x, hbits.y, hbits.z, hbits.w}, hs[4] = {hshift.x, hshift.y, hshift.z, hshift.w};
    uint32_t cnt = 0, rmin = 0, rw = 1;
    if (i < P) {
        const uint2 r = rect[i];
        rmin = r.x;
        rw = r.y & 0xFFFFu;
        cnt = rw * (r.y >> 16);
        if (rw == 0u) rw = 1u;
    }
    const bool heavy = gs2m_heavy(cnt, crowded);
    const uint32_t hu = heavy ? (cnt + GS2M_UNIT - 1u) / GS2M_UNIT : 0u;
    const uint32_t incl_all = wave_inclusive_scan_u32(cnt, lane), incl_hu = wave_inclusive_scan_u32(hu, lane);
    if (lane == 63) { s_wtot[wave] = incl_all; s_whu[wave] = incl_hu; }
    const uint32_t lcnt = heavy ? 0u : cnt;  // instances the wave expands itself
    if (lcnt > 0) {
        const float4* r = rec + (size_t)i * REC_Q;
        s_geo[wave][lane] = r[REC_GEO0];
        s_ct[wave][lane] = make_float2(r[REC_GEO1].x, r[REC_BIN].w);
        s_depth[wave][lane] = depth_key[i];
    }
    const uint32_t incl = wave_inclusive_scan_u32(lcnt, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    s_pref[wave][lane] = incl - lcnt;
    s_rmin[wave][lane] = rmin;
    s_rw[wave][lane] = rw;
    s_rc[wave][lane] = 0u;
    gs2m_sync();
    // ---- emission offsets: the exclusive prefix sum of tiles_touched in index order (the reference's InclusiveSum,
    // rasterizer_impl.cu:265-266) = the block's prefix + the waves in front + the lanes in front; the heavy units likewise
    uint32_t off = block_pref[blockIdx.x], ustart = block_hupref[blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; w++)
        if (w < wave) { off += s_wtot[w]; ustart += s_whu[w]; }
    off += incl_all - cnt;
    ustart += incl_hu - hu;
    s_off[wave][lane] = off;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) counters[0] = off + cnt;  // num_rendered (debug mode compares it with what the host was told)
    if (heavy) {
        gauss_rows[i] = GS2M_ROWS_BIG | ustart;
        for (uint32_t j = 0; j < hu; j++) {
            hrec[ustart + j].gid = (uint32_t)i;
            hrec[ustart + j].off = off;
        }
    }
    const uint32_t gid0 = (uint32_t)blockIdx.x * 256u;
    uint32_t rows_run = 0;  // gradient rows of the wave's own instances so far
    for (uint32_t k = 0; k < total; k += GS2M_WAVE) {
        const uint32_t j = k + lane;
        uint32_t pc = 0, slot = 0, mask = 0;
        int lo = 0;
        if (j < total) {
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (s_pref[wave][lo + step] <= j) lo += step;  // lo + step <= 63 always (a heavy Gaussian has an empty range: never found)
            const uint32_t t = j - s_pref[wave][lo];
            slot = s_off[wave][lo] + t;
            const uint2 km = expand_instance(s_rmin[wave][lo], s_rw[wave][lo], t, s_geo[wave][lo], s_ct[wave][lo], W, H, tiles_x);
            mask = km.y;
            keys_out[slot] = km.x;
            for (int p = 0; p < npass; p++) atomicAdd(&s_th[p][(km.x >> hs[p]) & ((1u << hb[p]) - 1u)], 1u);
            pc = (uint32_t)__popc(mask);
            if (pc) atomicAdd(&s_rc[wave][lo], pc);
        }
        const uint32_t pin = wave_inclusive_scan_u32(pc, lane);
        // value = id | mask, and the instance's first gradient row relative to the wave's first row
        if (j < total) e_rec[slot] = make_uint4((gid0 + (uint32_t)(wave * GS2M_WAVE + lo)) | (mask << GS2M_GID_BITS), rows_run + pin - pc, s_depth[wave][lo], 0u);
        rows_run += __shfl(pin, 63, 64);
    }
    if (i < P && !heavy) gauss_rows[i] = s_rc[wave][lane];  // LDS operations of one wave execute in order: the adds are done
    const size_t wave_id = (size_t)blockIdx.x * 4 + wave;
    if (lane == 0 && (size_t)wave_id * GS2M_WAVE < (size_t)P) wave_rows[wave_id] = rows_run;
    gs2m_sync();
    // this workgroup's digit counts into one of GS2M_HIST_COPIES copies of the global histogram (one add per non-empty bin)
    uint32_t* dst = tile_hist + (blockIdx.x & (GS2M_HIST_COPIES - 1)) * GS2M_HIST_COPY_WORDS;
    for (int p = 0; p < npass; p++) {
        const uint32_t c = s_th[p][threadIdx.x];
        if (c) atomicAdd(&dst[p * 256 + threadIdx.x], c);
    }
}

// block_hu once more, with the crowded-wave rule off (common.h: gs2m_heavy; api.hip launches this when the rule would reserve more rows
// than the frame can justify, then scans again)
__global__ void __launch_bounds__(256) recount_heavy_kernel(int P, const uint2* __restrict__ rect, uint32_t* __restrict__ block_hu) {
    __shared__ uint32_t s_hu[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t cnt = 0;
    if (i < P) {
        const uint2 r = rect[i];
        cnt = (r.y & 0xFFFFu) * (r.y >> 16);
    }
    uint32_t hu = gs2m_heavy(cnt, GS2M_CROWDED_OFF) ? (cnt + GS2M_UNIT - 1u) / GS2M_UNIT : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) hu += (uint32_t)__shfl_xor((int)hu, d, 64);
    if ((threadIdx.x & 63) == 0) s_hu[threadIdx.x >> 6] = hu;
    gs2m_sync();
    if (threadIdx.x == 0) block_hu[blockIdx.x] = s_hu[0] + s_hu[1] + s_hu[2] + s_hu[3];
}

// One wave per heavy unit: instances 64 (u - first unit) .. + 63 of the Gaussian the unit belongs to.  Slot = the Gaussian's first
// slot + the instance number (index order, as everything else); gradient row of the instance = GS2M_ROWS_BIG | 256 u + 4 lane (four
// rows reserved per instance; HeavyUnit::pop says how many are used).
__global__ void __launch_bounds__(256) emit_heavy_kernel(uint32_t units, int W, int H, int tiles_x, const uint2* __restrict__ rect, const float4* __restrict__ rec,
                                                         const uint32_t* __restrict__ depth_key, const uint32_t* __restrict__ gauss_rows,
                                                         uint32_t* __restrict__ keys_out, uint4* __restrict__ e_rec, HeavyUnit* __restrict__ hrec,
                                                         uint32_t* __restrict__ tile_hist, int npass, int4 hbits, int4 hshift) {
    __shared__ uint32_t s_th[4][256];
#pragma unroll
    for (int p = 0; p < 4; p++) s_th[p][threadIdx.x] = 0u;
    gs2m_sync();
    const int hb[4] = {hbits.x, hbits.y, hbits.z, hbits.w}, hs[4] = {hshift.x, hshift.y, hshift.z, hshift.w};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x * 4u + (uint32_t)wave;
    if (u < units) {
        const uint32_t gid = hrec[u].gid, off = hrec[u].off;  // (wave-uniform)
        const uint32_t ustart = gauss_rows[gid] & ~GS2M_ROWS_BIG;
        const uint2 r = rect[gid];
        const uint32_t rw = r.y & 0xFFFFu, cnt = rw * (r.y >> 16);
        const float4* rq = rec + (size_t)gid * REC_Q;
        const float4 geo = rq[REC_GEO0];
        const float2 ct = make_float2(rq[REC_GEO1].x, rq[REC_BIN].w);
        const uint32_t dk = depth_key[gid];
        const uint32_t t = (u - ustart) * GS2M_UNIT + (uint32_t)lane;
        uint32_t pc = 0;
        if (t < cnt) {
            const uint2 km = expand_instance(r.x, rw, t, geo, ct, W, H, tiles_x);
            keys_out[off + t] = km.x;
            for (int p = 0; p < npass; p++) atomicAdd(&s_th[p][(km.x >> hs[p]) & ((1u << hb[p]) - 1u)], 1u);
            e_rec[off + t] = make_uint4(gid | (km.y << GS2M_GID_BITS), GS2M_ROWS_BIG | (4u * (u * GS2M_UNIT + (uint32_t)lane)), dk, 0u);
            pc = (uint32_t)__popc(km.y);
        }
        hrec[u].pop[lane] = (uint8_t)pc;
    }
    gs2m_sync();
    uint32_t* dst = tile_hist + (blockIdx.x & (GS2M_HIST_COPIES - 1)) * GS2M_HIST_COPY_WORDS;
    for (int p = 0; p < npass; p++) {
        const uint32_t c = s_th[p][threadIdx.x];
        if (c) atomicAdd(&dst[p * 256 + threadIdx.x], c);
    }
}

}  // namespace

void gs2m_launch_blockscan(int P, const GeomState& g, uint32_t* landing, hipStream_t s) {
    const size_t nb = (size_t)(P + 255) / 256;
    blockscan_kernel<<<(unsigned)((nb + 1023) / 1024 > 0 ? (nb + 1023) / 1024 : 1), 1024, 0, s>>>(g.block_tt, g.block_hu, nb, g.block_pref, g.block_hupref, g.counters,
                                                                                                  landing);
}

void gs2m_launch_recount_heavy(int P, const GeomState& g, hipStream_t s) {
    recount_heavy_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, g.rect, g.block_hu);
}

void gs2m_launch_emit(int P, int W, int H, int tiles_x, int tile_bits, const GeomState& g, const BinningState& b, uint32_t heavy_units, uint32_t crowded,
                      uint32_t* landing, const ZeroJobs& zero, hipStream_t s) {
    int npass = 0, bits[4], shift[4];
    gs2m_radix_plan(tile_bits, &npass, bits, shift);  // the digits the tile sort will use: counted here, where the keys are made
    const int4 hbits = make_int4(bits[0], bits[1], bits[2], bits[3]), hshift = make_int4(shift[0], shift[1], shift[2], shift[3]);
    emit_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, W, H, tiles_x, g.rect, g.block_pref, g.block_hupref, g.rec, g.depth_key, b.keys_unsorted, b.e_rec, b.hrec,
                                                g.gauss_rows, g.wave_rows, g.counters, g.tile_hist, npass, hbits, hshift, crowded, zero);
    if (heavy_units > 0u)
        emit_heavy_kernel<<<(heavy_units + 3u) / 4u, 256, 0, s>>>(heavy_units, W, H, tiles_x, g.rect, g.rec, g.depth_key, g.gauss_rows, b.keys_unsorted, b.e_rec,
                                                                  b.hrec, g.tile_hist, npass, hbits, hshift);
    const size_t nw = (size_t)(P + 63) / 64;
    rowscan_kernel<<<(unsigned)((nw + 1023) / 1024 > 0 ? (nw + 1023) / 1024 : 1), 1024, 0, s>>>(g.wave_rows, nw, g.wave_rowbase, g.counters, landing);
}

// Zero fill as an ordinary kernel.  hipMemsetAsync goes through the runtime's blit path, which on this stack
// leaves a ~10 us bubble on the stream around every call (kernel traces: tools/trace_timeline.sh); six of them
// per view were 3 % of the step.  `bytes` must be a multiple of 4, `p` 4-byte aligned.
__global__ void zero_kernel(uint32_t* __restrict__ p, size_t words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t head = (4 - (((uintptr_t)p >> 2) & 3)) & 3;  // words up to the first 16-B boundary
    const size_t quads = words > head ? (words - head) >> 2 : 0;
    if (i < quads) reinterpret_cast<uint4*>(p + head)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (i < head && i < words) p[i] = 0u;
    const size_t tail = head + 4 * quads;
    if (i < words - min(words, tail)) p[tail + i] = 0u;
}
hipError_t gs2m_zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) || ((uintptr_t)p & 3)) return hipMemsetAsync(p, 0, bytes, s);
    const size_t words = bytes >> 2, threads = (words >> 2) + 4;
    zero_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((uint32_t*)p, words);
    return hipGetLastError();
}
