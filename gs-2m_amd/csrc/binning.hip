// Tile binning for gfx950 (round 5: no global sort).
//
// The reference (cuda_rasterizer/rasterizer_impl.cu:63-103, 265-305) scans tiles_touched, expands every visible Gaussian into
// (tile << 32 | depth) u64 keys and runs one 45-bit radix sort over all R instances, then finds the tile ranges in the sorted
// keys.  Rounds 1-4 here sorted the P Gaussians by depth (4 radix passes), emitted instances in depth order and sorted them
// stably by tile (2 passes): six dependent sort launches whose cost is look-back latency, not bytes.  Now the instances are
// BUCKETED by tile -- a counting sort whose counts, offsets and placement are three small kernels -- and each tile's span is
// ordered by (depth, Gaussian id) ON CHIP (tile_sort.hip):
//   count_kernel  per Gaussian (index order): one integer atomic per tile of its rectangle into tile_count[]; its first
//                 workgroup publishes num_rendered (the sum of the per-block counts the preprocess kernel left) to the host,
//                 which sizes the binning buffer while this kernel and the next run;
//   scan_kernel   two workgroups: tile_count -> ranges[] / cursor[] (identifyTileRanges, rasterizer_impl.cu:108-129, before
//                 anything is placed) and the longest list (selects the sort kernel); block_tt -> block_pref (the reference's
//                 InclusiveSum, at block granularity);
//   fill_kernel   load-balanced expansion as before (a wave owns 64 Gaussians and places 64 instances per step; Gaussians over
//                 hundreds of tiles are expanded by the whole workgroup): exact ellipse-vs-8x8 test per instance -> 4-bit
//                 quadrant mask; slot = atomic bump of the tile's cursor; {depth, id | mask, first gradient row} stored there;
//                 gradient rows numbered densely per wave in index order;
//   rowscan_kernel  exclusive prefix of the waves' row counts: rows are dense over the whole view (the backward's scratch is
//                 sized by their number, published to the host).
// The order inside a tile's span is the arrival order of the atomics -- arbitrary -- and irrelevant: the span is sorted by
// (depth, id) afterwards, which is exactly the reference's order within a tile (ties in id order).
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
    g.wave_rows = (uint32_t*)take(nw * 4);
    g.wave_rowbase = (uint32_t*)take(nw * 4);
    g.total_bytes = off + GS2M_ALIGN;
    return g;
}

BinningState gs2m_carve_binning(char* base, size_t R) {
    BinningState b;
    size_t off = base ? (gs2m_align_up((size_t)(uintptr_t)base) - (size_t)(uintptr_t)base) : 0;
    auto take = [&](size_t bytes) {
        char* p = (char*)((uintptr_t)base + off);
        off = gs2m_align_up(off + bytes);
        return p;
    };
    b.u_depth = (uint32_t*)take(R * 4);
    b.u_val = (uint32_t*)take(R * 4);
    b.u_row = (uint32_t*)take(R * 4);
    b.point_list = (uint32_t*)take(R * 4);
    b.tile_keys = (uint32_t*)take(R * 4);
    b.qlist = (uint2*)take(R * 4 * sizeof(uint2));
    b.qrow = (uint32_t*)take(R * 4 * 4);
    b.total_bytes = off + GS2M_ALIGN;
    return b;
}

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
    im.tile_count = (uint32_t*)take(tiles * 4);
    im.cursor = (uint32_t*)take(tiles * 4);
    im.qcount = (uint32_t*)take(tiles * 4 * sizeof(uint32_t));
    im.qlast = (uint32_t*)take(tiles * 4 * sizeof(uint32_t));
    im.total_bytes = off + GS2M_ALIGN;
    return im;
}

namespace {

__device__ __forceinline__ void publish(uint32_t* landing, int word, uint32_t v) {
    // one aligned system-scope 32-bit store into the mapped pinned block the host polls (a 4-byte hipMemcpyAsync may be
    // carried out byte by byte: torn counts were seen)
    __hip_atomic_store(landing + word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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

// ---- count: tile histogram --------------------------------------------------------------------------------------------
// One thread per Gaussian in index order, one fire-and-forget integer atomic per tile of its rectangle (8 bytes of input per
// Gaussian: the rectangle the preprocess kernel left in GeomState::rect).  Rectangles of GS2M_BIG_TILES tiles and more are
// walked by the whole wave afterwards.  Workgroup 0 first adds up the per-block instance counts and publishes num_rendered:
// the host sizes the binning buffer and queues the fill kernel while this kernel and the scan run.
__global__ void __launch_bounds__(256) count_kernel(int P, int tiles_x, const uint2* __restrict__ rect,
                                                    const uint32_t* __restrict__ block_tt, int nblocks,
                                                    uint32_t* __restrict__ tile_count, uint32_t* landing) {
    __shared__ unsigned long long s_part[16];
    if (blockIdx.x == 0) {
        unsigned long long t = 0;
        for (int b = threadIdx.x; b < nblocks; b += 256) t += block_tt[b];
        t = block_sum_u64(t, s_part);
        // saturated: a count beyond 2^32 cannot wrap past the caller's range check
        if (threadIdx.x == 0) publish(landing, GS2M_LAND_R, t > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)t);
    }
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    uint2 r = make_uint2(0u, 0u);
    if (i < P) r = rect[i];
    const uint32_t w = r.y & 0xFFFFu, h = r.y >> 16, cnt = w * h;
    const bool big = cnt >= GS2M_BIG_TILES;
    if (cnt != 0u && !big) {
        uint32_t t = (r.x >> 16) * (uint32_t)tiles_x + (r.x & 0xFFFFu), tx = 0;
        for (uint32_t k = 0; k < cnt; k++) {
            __hip_atomic_fetch_add(&tile_count[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t++;
            if (++tx == w) { tx = 0; t += (uint32_t)tiles_x - w; }
        }
    }
    unsigned long long m = __builtin_amdgcn_ballot_w64(big);
    while (m != 0ull) {  // a splat over hundreds of tiles: all 64 lanes on its rectangle
        const int src = __builtin_ctzll(m);
        m &= m - 1ull;
        const uint32_t rx = __shfl(r.x, src, 64), ry = __shfl(r.y, src, 64);
        const uint32_t bw = ry & 0xFFFFu, n = bw * (ry >> 16);
        const uint32_t t0 = (rx >> 16) * (uint32_t)tiles_x + (rx & 0xFFFFu);
        for (uint32_t k = lane; k < n; k += GS2M_WAVE) {
            const uint32_t yy = k / bw, xx = k - yy * bw;
            __hip_atomic_fetch_add(&tile_count[t0 + yy * (uint32_t)tiles_x + xx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- scan: tile ranges and block prefixes ------------------------------------------------------------------------------
// Exclusive prefix sum of in[0, n) by ONE workgroup of 1024 threads: a thread owns a contiguous segment (a serial sum, a
// workgroup-wide scan of the 1024 segment sums, a serial write-out).  The arrays are a few thousand words (8160 tiles at
// 1080p, 3907 blocks at 1M Gaussians): latency, not bandwidth.  `emit(i, exclusive, value)` receives every element.
template <typename F>
__device__ __forceinline__ unsigned long long scan_segments(const uint32_t* __restrict__ in, size_t n, uint32_t* s_w /* [16] */, F emit,
                                                            uint32_t* out_max) {
    const size_t seg = (n + 1023) / 1024, lo = min(n, (size_t)threadIdx.x * seg), hi = min(n, lo + seg);
    uint32_t sum = 0, mx = 0;
    for (size_t i = lo; i < hi; i++) {
        const uint32_t v = in[i];
        sum += v;
        mx = max(mx, v);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_scan_u32(sum, lane);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, d, 64));
    gs2m_sync();
    if (lane == 63) { s_w[wave] = incl; s_w[16 + wave] = mx; }
    gs2m_sync();
    uint32_t run = incl - sum, total = 0, gmax = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) run += s_w[w];
        total += s_w[w];
        gmax = max(gmax, s_w[16 + w]);
    }
    for (size_t i = lo; i < hi; i++) {
        const uint32_t v = in[i];
        emit(i, run, v);
        run += v;
    }
    if (out_max) *out_max = gmax;
    return total;
}

__global__ void __launch_bounds__(1024) scan_kernel(const uint32_t* __restrict__ tile_count, size_t tiles, uint2* __restrict__ ranges,
                                                    uint32_t* __restrict__ cursor, const uint32_t* __restrict__ block_tt, size_t nblocks,
                                                    uint32_t* __restrict__ block_pref, uint32_t* __restrict__ counters, uint32_t* landing) {
    __shared__ uint32_t s_w[32];
    if (blockIdx.x == 0) {
        uint32_t gmax = 0;
        scan_segments(tile_count, tiles, s_w, [&](size_t t, uint32_t excl, uint32_t v) {
            ranges[t] = v != 0u ? make_uint2(excl, excl + v) : make_uint2(0u, 0u);  // (0, 0) for an untouched tile, as the reference's memset leaves it
            cursor[t] = excl;
        }, &gmax);
        if (threadIdx.x == 0) publish(landing, GS2M_LAND_MAXTILE, gmax + 1u);  // + 1: 0 means "not landed"
    } else {
        const unsigned long long total = scan_segments(block_tt, nblocks, s_w, [&](size_t b, uint32_t excl, uint32_t) { block_pref[b] = excl; }, nullptr);
        if (threadIdx.x == 0) counters[1] = (uint32_t)total;
    }
}

// exclusive prefix of the waves' gradient-row counts -> first row of every wave; the total goes to the host
__global__ void __launch_bounds__(1024) rowscan_kernel(const uint32_t* __restrict__ wave_rows, size_t nwaves, uint32_t* __restrict__ wave_rowbase,
                                                       uint32_t* __restrict__ counters, uint32_t* landing) {
    __shared__ uint32_t s_w[32];
    const unsigned long long total = scan_segments(wave_rows, nwaves, s_w, [&](size_t w, uint32_t excl, uint32_t) { wave_rowbase[w] = excl; }, nullptr);
    if (threadIdx.x == 0) {
        counters[2] = (uint32_t)total;
        publish(landing, GS2M_LAND_ROWS, (uint32_t)total + 1u);
    }
}

// ---- fill: instances into their tiles' spans ---------------------------------------------------------------------------
// Load-balanced expansion (replaces duplicateWithKeys, rasterizer_impl.cu:63-103).  One wave owns 64 consecutive Gaussians;
// each step the wave places 64 instances, each lane locating its source Gaussian by binary search in the wave's prefix sums.
// Every instance is tested against the four 8x8 quadrants of its tile with the exact ellipse-vs-rectangle test (common.h) --
// here the Gaussian's geometry is loaded once per Gaussian -- and the 4-bit hit mask travels above the Gaussian id.  The
// backward writes one gradient row per set bit; the rows of a wave's 64 Gaussians are numbered densely in emission order
// (Gaussian by Gaussian, instance by instance, quadrant by quadrant), relative to the wave's first row, which rowscan_kernel
// supplies afterwards (wave_rowbase) and tile_sort.hip adds: all rows of a Gaussian are one dense run and the per-Gaussian
// backward streams them (gaussian_bwd.hip).
// BIG SPLATS.  Gaussians with at least GS2M_BIG_TILES tiles are left out of the wave's own loop and expanded afterwards by
// ALL FOUR waves of the workgroup together (64-instance chunks dealt round the waves: a first pass counts the rows per chunk,
// the chunk totals are scanned in LDS, a second pass repeats the tests and places the instances).  Their rows follow the
// wave's small rows -- small Gaussians in lane order, then the big ones in lane order -- and gauss_rows carries
// GS2M_ROWS_BIG for them, which the per-Gaussian backward reads the same way.
__global__ void __launch_bounds__(256) fill_kernel(int P, int W, int H, int tiles_x, const uint2* __restrict__ rect,
                                                   const uint32_t* __restrict__ depth_key, const uint32_t* __restrict__ block_pref,
                                                   const float4* __restrict__ rec, uint32_t* __restrict__ cursor,
                                                   uint32_t* __restrict__ u_depth, uint32_t* __restrict__ u_val, uint32_t* __restrict__ u_row,
                                                   uint32_t* __restrict__ gauss_rows, uint32_t* __restrict__ wave_rows,
                                                   uint32_t* __restrict__ counters) {
    __shared__ uint32_t s_pref[4][GS2M_WAVE];
    __shared__ uint32_t s_rmin[4][GS2M_WAVE];
    __shared__ uint32_t s_rw[4][GS2M_WAVE];
    __shared__ uint32_t s_cnt[4][GS2M_WAVE];    // instances of the Gaussian
    __shared__ uint32_t s_depth[4][GS2M_WAVE];
    __shared__ float4 s_geo[4][GS2M_WAVE];      // x, y, A, B
    __shared__ float2 s_ct[4][GS2M_WAVE];       // C, t2
    __shared__ uint32_t s_rc[4][GS2M_WAVE];     // gradient rows per Gaussian
    __shared__ unsigned long long s_bigmask[4];  // per wave: lanes whose Gaussian is big
    __shared__ uint32_t s_smallrows[4], s_wtot[4];
    __shared__ uint32_t s_ctot[1024];  // rows per 64-instance chunk of the big Gaussian being expanded, then their exclusive prefix
    __shared__ uint32_t s_round_total;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t cnt = 0, rmin = 0, rw = 1;
    if (i < P) {
        const uint2 r = rect[i];
        rmin = r.x;
        rw = r.y & 0xFFFFu;
        cnt = rw * (r.y >> 16);
        if (rw == 0u) rw = 1u;
    }
    const uint32_t incl_all = wave_inclusive_scan_u32(cnt, lane);
    if (lane == 63) s_wtot[wave] = incl_all;
    if (cnt > 0) {
        const float4* r = rec + (size_t)i * REC_Q;
        s_geo[wave][lane] = r[REC_GEO0];
        s_ct[wave][lane] = make_float2(r[REC_GEO1].x, r[REC_BIN].w);
        s_depth[wave][lane] = depth_key[i];
    }
    const bool big = cnt >= GS2M_BIG_TILES && cnt < (1u << 29);  // (4 rows per instance at most: the row count must stay below the GS2M_ROWS_BIG bit)
    const uint32_t lcnt = big ? 0u : cnt;  // instances the wave expands itself
    const uint32_t incl = wave_inclusive_scan_u32(lcnt, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    s_pref[wave][lane] = incl - lcnt;
    s_rmin[wave][lane] = rmin;
    s_rw[wave][lane] = rw;
    s_cnt[wave][lane] = cnt;
    s_rc[wave][lane] = 0u;
    const unsigned long long bigmask = __builtin_amdgcn_ballot_w64(big);
    if (lane == 0) s_bigmask[wave] = bigmask;
    gs2m_sync();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)  // num_rendered as the offsets add up (debug mode compares it with what the host was told)
        counters[0] = block_pref[blockIdx.x] + s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
    const uint32_t gid0 = (uint32_t)blockIdx.x * 256u;
    // tile and quadrant-hit mask of instance t of the Gaussian parked at [w][lo]
    auto expand = [&](int w, int lo, uint32_t t, uint32_t& tile) -> uint32_t {
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
        tile = ty * (uint32_t)tiles_x + tx;
        return mask;
    };
    // the instance takes the next free slot of its tile's span
    auto place = [&](int w, int lo, uint32_t tile, uint32_t mask, uint32_t row) {
        const uint32_t slot = __hip_atomic_fetch_add(&cursor[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u_depth[slot] = s_depth[w][lo];
        u_val[slot] = (gid0 + (uint32_t)(w * GS2M_WAVE + lo)) | (mask << GS2M_GID_BITS);
        u_row[slot] = row;
    };
    uint32_t rows_run = 0;  // gradient rows of the wave's own (small) instances so far
    for (uint32_t k = 0; k < total; k += GS2M_WAVE) {
        const uint32_t j = k + lane;
        uint32_t pc = 0, tile = 0, mask = 0;
        int lo = 0;
        if (j < total) {
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (s_pref[wave][lo + step] <= j) lo += step;  // lo + step <= 63 always (a big Gaussian has an empty range: never found)
            mask = expand(wave, lo, j - s_pref[wave][lo], tile);
            pc = (uint32_t)__popc(mask);
            if (pc) atomicAdd(&s_rc[wave][lo], pc);
        }
        const uint32_t pin = wave_inclusive_scan_u32(pc, lane);
        if (j < total) place(wave, lo, tile, mask, rows_run + pin - pc);  // the instance's first gradient row, relative to the wave's first
        rows_run += __shfl(pin, 63, 64);
    }
    if (i < P && !big) gauss_rows[i] = s_rc[wave][lane];  // LDS operations of one wave execute in order: the adds are done
    if (lane == 0) s_smallrows[wave] = rows_run;
    const size_t wave_id = (size_t)blockIdx.x * 4 + wave;
    if (gs2m_sync_or(bigmask != 0ull) == 0) {  // no big Gaussian in this workgroup (the common case)
        if (lane == 0 && (size_t)wave_id * GS2M_WAVE < (size_t)P) wave_rows[wave_id] = rows_run;
        return;
    }
    // ---- the workgroup's big Gaussians, one after the other, all four waves on each ----
    for (int w = 0; w < 4; w++) {
        unsigned long long m = s_bigmask[w];
        uint32_t bigrows = 0;  // rows of wave w's earlier big Gaussians
        while (m != 0ull) {
            const int lo = __builtin_ctzll(m);
            m &= m - 1ull;
            const uint32_t bcnt = s_cnt[w][lo];
            const uint32_t first_row = s_smallrows[w] + bigrows;
            const uint32_t chunks = (bcnt + GS2M_WAVE - 1) / GS2M_WAVE;
            uint32_t done_rows = 0;  // rows of the rounds before this one
            for (uint32_t c0 = 0; c0 < chunks; c0 += 1024) {  // rounds of at most 1024 chunks (s_ctot)
                const uint32_t c1 = min(chunks, c0 + 1024u);
                for (uint32_t c = c0 + wave; c < c1; c += 4) {  // first pass: rows per chunk
                    const uint32_t t = c * GS2M_WAVE + lane;
                    uint32_t tile;
                    uint32_t pc = t < bcnt ? (uint32_t)__popc(expand(w, lo, t, tile)) : 0u;
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
                for (uint32_t c = c0 + wave; c < c1; c += 4) {  // second pass: the tests again, placement with the row numbers
                    const uint32_t t = c * GS2M_WAVE + lane;
                    uint32_t tile = 0, mask = 0;
                    if (t < bcnt) mask = expand(w, lo, t, tile);
                    const uint32_t pc = (uint32_t)__popc(mask);
                    const uint32_t pin = wave_inclusive_scan_u32(pc, lane);
                    if (t < bcnt) place(w, lo, tile, mask, first_row + done_rows + s_ctot[c - c0] + pin - pc);
                }
                done_rows += s_round_total;
                gs2m_sync();  // s_ctot is rewritten by the next round / the next Gaussian
            }
            if (threadIdx.x == 0) gauss_rows[blockIdx.x * 256 + w * GS2M_WAVE + lo] = done_rows | GS2M_ROWS_BIG;
            bigrows += done_rows;
        }
        if (threadIdx.x == 0 && ((size_t)blockIdx.x * 4 + w) * GS2M_WAVE < (size_t)P) wave_rows[(size_t)blockIdx.x * 4 + w] = s_smallrows[w] + bigrows;
    }
}

}  // namespace

void gs2m_launch_count(int P, int tiles_x, const GeomState& g, const ImageState& im, uint32_t* landing, hipStream_t s) {
    count_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, tiles_x, g.rect, g.block_tt, (P + 255) / 256, im.tile_count, landing);
}

void gs2m_launch_scan(int P, size_t tiles, const GeomState& g, const ImageState& im, uint32_t* landing, hipStream_t s) {
    scan_kernel<<<2, 1024, 0, s>>>(im.tile_count, tiles, im.ranges, im.cursor, g.block_tt, (size_t)(P + 255) / 256, g.block_pref, g.counters, landing);
}

void gs2m_launch_fill(int P, int W, int H, int tiles_x, const GeomState& g, const BinningState& b, const ImageState& im, uint32_t* landing,
                      hipStream_t s) {
    fill_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, W, H, tiles_x, g.rect, g.depth_key, g.block_pref, g.rec, im.cursor, b.u_depth, b.u_val, b.u_row,
                                                g.gauss_rows, g.wave_rows, g.counters);
    rowscan_kernel<<<1, 1024, 0, s>>>(g.wave_rows, (size_t)(P + 63) / 64, g.wave_rowbase, g.counters, landing);
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
