// Tile binning for gfx950.
//
// The reference (cuda_rasterizer/rasterizer_impl.cu:63-103, 265-305) expands every visible
// Gaussian into (tile<<32 | depth) u64 keys and runs one 45-bit radix sort over all R
// instances.  Here the work is split so that far fewer bytes move through HBM:
//   1. depth sort of the P Gaussians themselves (u32 fp32-bit key, id payload), stable; its last pass leaves the sums of
//      tiles_touched over blocks of 256 sorted Gaussians (radix_sort.hip: SideBuckets);
//   2. load-balanced emit: emission offsets from those block sums + a scan of the workgroup's own 256 counts (the
//      reference's InclusiveSum, no kernel of its own since round 4); instance (tile id, Gaussian id) pairs in depth order,
//      every wave writes 64 consecutive slots per step (the reference loops serially per thread); the digits of the keys
//      are counted here for the tile sort; big Gaussians are expanded by the whole workgroup;
//   3. stable sort of the R pairs on the tile bits only (13 bits at 1080p instead of 45); its last pass records every
//      tile's range (identifyTileRanges);
//   4. per tile: the sorted list -> four quadrant lists (quad_lists_kernel).
// Stable sort by tile of a depth-ordered list == sort by (tile, depth) with ties kept in
// Gaussian-id order, i.e. exactly the reference's sorted list.
// Sort primitives: radix_sort.hip (hand-written onesweep radix sort).
#include "common.h"

GeomState gs2m_carve_geom(char* base, size_t P, size_t temp_bytes) {
    GeomState g;
    size_t off = base ? (gs2m_align_up((size_t)(uintptr_t)base) - (size_t)(uintptr_t)base) : 0;
    auto take = [&](size_t bytes) {  // base == nullptr: the returned "pointers" are byte offsets
        char* p = (char*)((uintptr_t)base + off);
        off = gs2m_align_up(off + bytes);
        return p;
    };
    g.rec = (float4*)take(P * REC_Q * sizeof(float4));
    g.tiles_touched = (uint32_t*)take(P * 4);
    g.depth_key = (uint32_t*)take(P * 4);
    g.sort_keyA = (uint32_t*)take(P * 4);
    g.sort_valA = (uint32_t*)take(P * 4);
    g.depth_key_sorted = (uint32_t*)take(P * 4);
    g.sorted_gid = (uint32_t*)take(P * 4);
    g.sorted_off = (uint32_t*)take(P * 4);
    g.clamped = (uint8_t*)take(P);
    g.sh_dir = (float*)take(P * 9 * sizeof(float));
    g.counters = (uint32_t*)take(64 * 4);
    g.sorted_rows = (uint32_t*)take(P * 4);
    g.temp = take(temp_bytes);
    g.temp_bytes = temp_bytes;
    g.total_bytes = off + GS2M_ALIGN;
    return g;
}

BinningState gs2m_carve_binning(char* base, size_t R, size_t temp_bytes) {
    BinningState b;
    size_t off = base ? (gs2m_align_up((size_t)(uintptr_t)base) - (size_t)(uintptr_t)base) : 0;
    auto take = [&](size_t bytes) {  // base == nullptr: the returned "pointers" are byte offsets
        char* p = (char*)((uintptr_t)base + off);
        off = gs2m_align_up(off + bytes);
        return p;
    };
    b.keys_unsorted = (uint32_t*)take(R * 4);
    b.vals_unsorted = (uint32_t*)take(R * 4);
    b.sort_keyA = (uint32_t*)take(R * 4);
    b.sort_valA = (uint32_t*)take(R * 4);
    b.tile_keys = (uint32_t*)take(R * 4);
    b.point_list = (uint32_t*)take(R * 4);
    b.inst_obs = (uint32_t*)take(R * 4);
    b.qlist = (uint2*)take(R * 4 * sizeof(uint2));
    b.temp = take(temp_bytes);
    b.temp_bytes = temp_bytes;
    b.total_bytes = off + GS2M_ALIGN;
    return b;
}

ImageState gs2m_carve_image(char* base, size_t N, size_t tiles) {
    ImageState im;
    size_t off = base ? (gs2m_align_up((size_t)(uintptr_t)base) - (size_t)(uintptr_t)base) : 0;
    auto take = [&](size_t bytes) {  // base == nullptr: the returned "pointers" are byte offsets
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

size_t gs2m_geom_temp_bytes(size_t P) {
    // the depth sort's and the scan's scratch side by side: the preprocess kernel zeroes both ahead of time
    return gs2m_align_up(gs2m_radix_temp_bytes(P, 32)) + gs2m_align_up(gs2m_front_temp_bytes(P)) + GS2M_ALIGN;
}

size_t gs2m_binning_temp_bytes(size_t R, int tile_bits) { return gs2m_align_up(gs2m_radix_temp_bytes(R, tile_bits)) + GS2M_ALIGN; }

namespace {

// Load-balanced expansion (replaces duplicateWithKeys, rasterizer_impl.cu:63-103).
// One wave owns 64 depth-sorted Gaussians whose instances occupy one contiguous slot range;
// each step the wave writes 64 consecutive slots, each lane locating its source Gaussian by
// binary search in the wave's prefix sums.  Also stores the emission offset into the record.
// Every instance is also tested against the four 8x8 quadrants of its tile with the
// exact ellipse-vs-rectangle test (common.h) -- here the Gaussian's geometry is loaded once per Gaussian, the second
// binning level (quad_lists_kernel) then needs no record gather at all -- and the 4-bit hit mask travels through the
// tile sort above the Gaussian id.  The backward writes one gradient row per set bit; the rows of a wave's 64
// Gaussians are numbered densely in emission order (instance by instance, quadrant by quadrant): an instance gets its
// first row here; the range of wave w starts at row 4 x (emission offset of its first Gaussian)
// -- a wave with n instances owns at most 4 n rows, so the ranges cannot overlap and need no prefix sum over the waves
// (the row scratch is sized for 4 R rows anyway).  So the rows of every Gaussian are one dense run and the
// per-Gaussian sum streams them (gaussian_bwd.hip).
// BIG SPLATS (round 4).  A Gaussian over hundreds of tiles (a close-up, a background blob) used to be walked by its one wave,
// 64 tiles per step, while the wave's other 63 Gaussians waited: a thousand such splats doubled this kernel's time.  Gaussians
// with at least GS2M_BIG_TILES tiles are now left out of the wave's own loop and expanded afterwards by ALL FOUR waves of the
// workgroup together (64-instance chunks dealt round the waves: tests and key / value stores in a first pass, chunk totals
// scanned in LDS, first-row numbers in a second pass that reads the masks back).  Their rows follow the wave's small rows
// inside the wave's range -- small Gaussians in lane order from row 4 x (first slot), then the big ones in lane order --
// and sorted_rows carries GS2M_ROWS_BIG for them, which row_reduce_dense_kernel (gaussian_bwd.hip) reads the same way.
__global__ void __launch_bounds__(256) emit_kernel(int P, int W, int H, int tiles_x, const uint32_t* __restrict__ sorted_gid,
                                                   const uint32_t* __restrict__ tiles_touched,
                                                   const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ super_sums,
                                                   uint32_t* __restrict__ counters,
                                                   uint32_t* __restrict__ sorted_off, float4* __restrict__ rec,
                                                   uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                   uint32_t* __restrict__ inst_obs,
                                                   uint32_t* __restrict__ sorted_rows, uint32_t* __restrict__ tile_hist,
                                                   int npass, int4 hbits, int4 hshift, ZeroJobs zero) {
    __shared__ uint32_t s_th[4][256];  // digit counts of this workgroup's keys, for the tile sort (radix_sort.hip: ext_hist)
    __shared__ uint32_t s_pref[4][GS2M_WAVE];
    __shared__ uint32_t s_gid[4][GS2M_WAVE];
    __shared__ uint32_t s_rmin[4][GS2M_WAVE];
    __shared__ uint32_t s_rw[4][GS2M_WAVE];
    __shared__ uint32_t s_off[4][GS2M_WAVE];  // first emission slot of the Gaussian
    __shared__ uint32_t s_cnt[4][GS2M_WAVE];  // its instances
    __shared__ float4 s_geo[4][GS2M_WAVE];   // x, y, A, B
    __shared__ float2 s_ct[4][GS2M_WAVE];    // C, t2
    __shared__ uint32_t s_rc[4][GS2M_WAVE];  // gradient rows per Gaussian
    __shared__ unsigned long long s_bigmask[4];  // per wave: lanes whose Gaussian is big
    __shared__ uint32_t s_base4[4], s_smallrows[4], s_wsum[4], s_wtot[4];
    __shared__ uint32_t s_ctot[1024];  // rows per 64-instance chunk of the big Gaussian being expanded, then their exclusive prefix
    __shared__ uint32_t s_round_total;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    gs2m_zero_jobs(zero, (size_t)i, (size_t)gridDim.x * 256);  // tile-sort scratch and the tile ranges
#pragma unroll
    for (int p = 0; p < 4; p++) s_th[p][threadIdx.x] = 0u;
    const int hb[4] = {hbits.x, hbits.y, hbits.z, hbits.w}, hs[4] = {hshift.x, hshift.y, hshift.z, hshift.w};
    // this workgroup's counts into one of GS2M_HIST_COPIES copies of the global histogram (one add per non-empty bin)
    auto flush_hist = [&]() {
        uint32_t* dst = tile_hist + (blockIdx.x & (GS2M_HIST_COPIES - 1)) * GS2M_HIST_COPY_WORDS;
        for (int p = 0; p < npass; p++) {
            const uint32_t c = s_th[p][threadIdx.x];
            if (c) atomicAdd(&dst[p * 256 + threadIdx.x], c);
        }
    };
    uint32_t cnt = 0, gid = 0, rmin = 0, rw = 1;
    if (i < P) {
        gid = sorted_gid[i];
        cnt = tiles_touched[gid];
    }
    // ---- emission offsets: the exclusive prefix sum of tiles_touched in depth order (the reference's InclusiveSum,
    // rasterizer_impl.cu:265-266).  The sum over all workgroups in front comes from the block sums the depth sort's last pass
    // left behind (one per 256 Gaussians = one per workgroup of this kernel; radix_sort.hip: SideBuckets) -- every thread adds
    // its share of them, no chain between workgroups -- the rest is a scan of the workgroup's own 256 counts.  Rounds 1-3
    // ran a scan kernel (a gather, a look-back chain, two P-sized arrays) in front of this one.
    // two levels, so that every thread adds one block sum (the blocks of this workgroup's own super-block of 256) and --
    // beyond 16 M Gaussians: a few -- super-block sums, all requested at once: one memory round trip
    uint32_t bsum = 0;
    {
        const uint32_t mysuper = blockIdx.x >> 8, b = (mysuper << 8) + threadIdx.x;
        if (b < blockIdx.x) bsum = block_sums[b];
        for (uint32_t sp = threadIdx.x; sp < mysuper; sp += 256) bsum += super_sums[sp];
    }
    bsum = wave_inclusive_scan_u32(bsum, lane);
    const uint32_t incl_all = wave_inclusive_scan_u32(cnt, lane);
    if (lane == 63) { s_wsum[wave] = bsum; s_wtot[wave] = incl_all; }
    gs2m_sync();
    uint32_t off = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];  // first emission slot of the workgroup ...
#pragma unroll
    for (int w = 0; w < 4; w++)
        if (w < wave) off += s_wtot[w];
    const uint32_t base = off;  // ... of the wave ...
    off += incl_all - cnt;      // ... of the Gaussian
    if (i < P) sorted_off[i] = off;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) counters[0] = off + cnt;  // num_rendered (debug mode checks the side sum against it)
    if (cnt > 0) {
        float4* r = rec + (size_t)gid * REC_Q + REC_BIN;
        const float4 bin = *r;
        rmin = f2u(bin.y);
        rw = f2u(bin.z) & 0xFFFFu;
        // the Gaussian's first emission slot, for the backward's row lookup: the ONE scattered word this kernel writes into
        // the records (a compact by-id array instead saved 1.6 us here and cost the backward blend 0.25 GB of sector
        // traffic for its gathers; the second word rounds 2-3 wrote, the wave's first row, is folded into inst_obs below)
        reinterpret_cast<uint32_t*>(r)[0] = off;
        s_geo[wave][lane] = rec[(size_t)gid * REC_Q + REC_GEO0];
        s_ct[wave][lane] = make_float2(rec[(size_t)gid * REC_Q + REC_GEO1].x, bin.w);
    }
    const bool big = cnt >= GS2M_BIG_TILES && cnt < (1u << 29);  // (4 rows per instance at most: the row count must stay below the GS2M_ROWS_BIG bit)
    const uint32_t lcnt = big ? 0u : cnt;  // instances the wave expands itself
    const uint32_t incl = wave_inclusive_scan_u32(lcnt, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    s_pref[wave][lane] = incl - lcnt;
    s_gid[wave][lane] = gid;
    s_rmin[wave][lane] = rmin;
    s_rw[wave][lane] = rw;
    s_off[wave][lane] = off;
    s_cnt[wave][lane] = cnt;
    s_rc[wave][lane] = 0u;
    const unsigned long long bigmask = __builtin_amdgcn_ballot_w64(big);
    if (lane == 0) { s_bigmask[wave] = bigmask; s_base4[wave] = base; }
    gs2m_sync();
    // tile and quadrant-hit mask of instance t of the Gaussian parked at [w][lo]; writes its key / value at `slot`
    auto expand = [&](int w, int lo, uint32_t t, uint32_t slot) -> uint32_t {
        const uint32_t rwid = s_rw[w][lo];
        const uint32_t ry = t / rwid, rx = t - ry * rwid;
        const uint32_t rm = s_rmin[w][lo];
        const uint32_t tx = (rm & 0xFFFFu) + rx, ty = (rm >> 16) + ry;
        const float4 a = s_geo[w][lo];
        const float2 ct = s_ct[w][lo];
        const int px0 = (int)tx * GS2M_TILE, py0 = (int)ty * GS2M_TILE;
        uint32_t mask = gs2m_reaches_quads(a.x, a.y, a.z, a.w, ct.x, ct.y, (float)px0, (float)py0);
        // quadrants outside the image have no pixels: no list entry, no gradient row
        if (px0 + 8 >= W) mask &= 0x5u;
        if (py0 + 8 >= H) mask &= 0x3u;
        const uint32_t key = ty * (uint32_t)tiles_x + tx;
        keys_out[slot] = key;
        vals_out[slot] = s_gid[w][lo] | (mask << GS2M_GID_BITS);
        for (int p = 0; p < npass; p++) atomicAdd(&s_th[p][(key >> hs[p]) & ((1u << hb[p]) - 1u)], 1u);
        return mask;
    };
    uint32_t rows_run = 0;  // gradient rows of the wave's own (small) instances so far
    for (uint32_t k = 0; k < total; k += GS2M_WAVE) {
        const uint32_t j = k + lane;
        uint32_t pc = 0, slot = 0;
        if (j < total) {
            int lo = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (s_pref[wave][lo + step] <= j) lo += step;  // lo + step <= 63 always (a big Gaussian has an empty range: never found)
            const uint32_t t = j - s_pref[wave][lo];
            slot = s_off[wave][lo] + t;
            pc = (uint32_t)__popc(expand(wave, lo, t, slot));
            if (pc) atomicAdd(&s_rc[wave][lo], pc);
        }
        {
            const uint32_t pin = wave_inclusive_scan_u32(pc, lane);
            // the instance's first gradient row: the wave's range starts at row 4 x (its first emission slot)
            if (j < total) inst_obs[slot] = 4u * base + rows_run + pin - pc;
            rows_run += __shfl(pin, 63, 64);
        }
    }
    if (i < P && !big) sorted_rows[i] = s_rc[wave][lane];  // LDS operations of one wave execute in order: the adds are done
    if (lane == 0) s_smallrows[wave] = rows_run;
    if (gs2m_sync_or(bigmask != 0ull) == 0) {  // no big Gaussian in this workgroup (the common case)
        flush_hist();
        return;
    }
    // ---- the workgroup's big Gaussians, one after the other, all four waves on each ----
    for (int w = 0; w < 4; w++) {
        unsigned long long m = s_bigmask[w];
        uint32_t bigrows = 0;  // rows of wave w's earlier big Gaussians
        while (m != 0ull) {
            const int lo = __builtin_ctzll(m);
            m &= m - 1ull;
            const uint32_t bcnt = s_cnt[w][lo], boff = s_off[w][lo];
            const uint32_t first_row = 4u * s_base4[w] + s_smallrows[w] + bigrows;
            const uint32_t chunks = (bcnt + GS2M_WAVE - 1) / GS2M_WAVE;
            uint32_t done_rows = 0;  // rows of the rounds before this one
            for (uint32_t c0 = 0; c0 < chunks; c0 += 1024) {  // rounds of at most 1024 chunks (s_ctot)
                const uint32_t c1 = min(chunks, c0 + 1024u);
                for (uint32_t c = c0 + wave; c < c1; c += 4) {  // first pass: tests, keys and values, rows per chunk
                    const uint32_t t = c * GS2M_WAVE + lane;
                    uint32_t pc = t < bcnt ? (uint32_t)__popc(expand(w, lo, t, boff + t)) : 0u;
                    pc = wave_inclusive_scan_u32(pc, lane);
                    if (lane == 63) s_ctot[c - c0] = pc;
                }
                gs2m_sync();
                if (wave == 0) {  // exclusive prefix over the round's chunk totals, 16 per lane
                    uint32_t v[16], sum = 0;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const uint32_t idx = lane * 16 + e;
                        v[e] = idx < c1 - c0 ? s_ctot[idx] : 0u;
                        sum += v[e];
                    }
                    const uint32_t inc = wave_inclusive_scan_u32(sum, lane);
                    uint32_t run = inc - sum;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const uint32_t idx = lane * 16 + e;
                        if (idx < c1 - c0) s_ctot[idx] = run;
                        run += v[e];
                    }
                    if (lane == 63) s_round_total = inc;
                }
                gs2m_sync();
                for (uint32_t c = c0 + wave; c < c1; c += 4) {  // second pass: first rows (the masks are read back: this thread wrote them)
                    const uint32_t t = c * GS2M_WAVE + lane;
                    const uint32_t pc = t < bcnt ? (uint32_t)__popc(vals_out[boff + t] >> GS2M_GID_BITS) : 0u;
                    const uint32_t pin = wave_inclusive_scan_u32(pc, lane);
                    if (t < bcnt) inst_obs[boff + t] = first_row + done_rows + s_ctot[c - c0] + pin - pc;
                }
                done_rows += s_round_total;
                gs2m_sync();  // s_ctot is rewritten by the next round / the next Gaussian
            }
            if (threadIdx.x == 0) sorted_rows[blockIdx.x * 256 + w * GS2M_WAVE + lo] = done_rows | GS2M_ROWS_BIG;
            bigrows += done_rows;
        }
    }
    flush_hist();  // (a barrier closes the last round above: every count is in)
}

// Second binning level: the sorted list of a 16x16 tile -> four order-preserving lists, one per 8x8 quadrant,
// holding only the instances that can reach the quadrant with alpha >= 1/255 (the exact ellipse-vs-rectangle
// test of common.h, evaluated by emit_kernel<true>, whose 4-bit result arrives above the Gaussian id in the sorted
// values; dropped instances contribute to no pixel of the quadrant, so every output is unchanged).
// The blend kernels then run one wave per quadrant straight down its list: no staging of instances that
// are skipped anyway, no tests, no ballot walks.  Entries keep the position in the tile list, so n_contrib
// (a tile-list position, as in the reference) and the gradient-row addressing stay what they were.
// Also identifyTileRanges (rasterizer_impl.cu:108-129): the tile sort's last pass has recorded where every tile's run of
// instances starts and ends (radix_sort.hip: range_raw); this kernel writes ranges[tile] from it ((0, 0) for an untouched tile,
// as the reference's memset leaves it) -- rounds 1-3 ran a kernel over all R sorted keys for that.
__global__ void __launch_bounds__(64) quad_lists_kernel(const uint32_t* __restrict__ ranges_raw, uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ point_list,
                                                        uint2* __restrict__ qlist, uint32_t* __restrict__ qcount) {
    // ONE WAVE per tile, 384 instances per step (six per lane, requested together): the 8160 tiles of a 1080p frame are one
    // generation of waves on the chip, with no LDS and no barrier -- ballots give every instance its place in each of the four
    // lists.  (Rounds 2-4 ran a workgroup of 256 threads per tile with the wave counts exchanged through LDS: four generations
    // of workgroups, 17 us; a tile list of a few hundred entries is one or two steps of a single wave.)
    constexpr int U = 6;
    const int tile = blockIdx.x, lane = threadIdx.x;
    const uint2 raw = reinterpret_cast<const uint2*>(ranges_raw)[tile];
    const uint2 range = raw.y != 0u ? make_uint2(~raw.x, raw.y) : make_uint2(0u, 0u);
    if (lane == 0) ranges[tile] = range;
    const int len = (int)(range.y - range.x);
    uint2* out = qlist + (size_t)4 * range.x;
    uint32_t run[4] = {0u, 0u, 0u, 0u};
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int base = 0; base < len; base += 64 * U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = base + u * 64 + lane;
            v[u] = k < len ? point_list[range.x + k] : 0u;  // Gaussian id | quadrant-hit mask << 28 (emit_kernel)
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = base + u * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool hit = ((v[u] >> (GS2M_GID_BITS + q)) & 1u) != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (hit) out[(size_t)q * len + run[q] + (uint32_t)__popcll(m & lt)] = make_uint2(v[u], (uint32_t)k);
                run[q] += (uint32_t)__popcll(m);
            }
        }
    }
    if (lane < 4) qcount[tile * 4 + lane] = lane == 0 ? run[0] : (lane == 1 ? run[1] : (lane == 2 ? run[2] : run[3]));
}

}  // namespace

void gs2m_launch_emit(int P, int W, int H, int tiles_x, int tile_bits, uint32_t* tile_hist, const uint32_t* block_sums, const uint32_t* super_sums,
                      const GeomState& g, const BinningState& b,
                      const ZeroJobs& zero, hipStream_t s) {
    int npass = 0, bits[4], shift[4];
    gs2m_radix_plan(tile_bits, &npass, bits, shift);  // the digits the tile sort will use: counted here, where the keys are made
    emit_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, W, H, tiles_x, g.sorted_gid, g.tiles_touched, block_sums, super_sums, g.counters, g.sorted_off, g.rec, b.keys_unsorted,
                                                b.vals_unsorted, b.inst_obs, g.sorted_rows, tile_hist, npass,
                                                make_int4(bits[0], bits[1], bits[2], bits[3]), make_int4(shift[0], shift[1], shift[2], shift[3]), zero);
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

void gs2m_launch_quad_lists(int W, int H, int tiles_x, int tiles_y, const GeomState& g, const BinningState& b,
                            const ImageState& im, hipStream_t s) {
    (void)W; (void)H; (void)g;
    quad_lists_kernel<<<tiles_x * tiles_y, 64, 0, s>>>(im.ranges_raw, im.ranges, b.point_list, b.qlist, im.qcount);
}
