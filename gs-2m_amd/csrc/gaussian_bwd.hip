// Per-Gaussian backward for gfx950: ONE kernel (round 5: the row sum was a kernel of its own while the rows lay in depth
// order and the tensors in index order; both are in index order now) that (1) sums the Gaussian's contiguous
// partial-gradient rows written by blend_bwd_q.hip, (2) back-propagates conic -> 2D covariance ->
// 3D covariance and view-space mean, (3) the projection of the 2D mean, (4) SH colour and
// (5) scale / rotation, and writes EVERY element of every gradient tensor (zeros for culled
// Gaussians), so the caller does not have to pre-zero 344 B per Gaussian as the reference's
// binding does (rasterize_points.cu:150-159).
//
// Semantics: computeCov2DCUDA backward.cu:153-281, preprocessCUDA (bwd) backward.cu:352-410,
// computeColorFromSH (bwd) backward.cu:23-148, computeCov3D (bwd) backward.cu:285-347 of
// diff-gaussian-rasterization/cuda_rasterizer, including the quirks listed in SURVEY.md A.5
// (+0.3 low-pass only here, clamp masks, no quaternion normalisation).
#include "common.h"

namespace {

struct M3 {
    float m[3][3];  // m[col][row]
};
__device__ __forceinline__ M3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    M3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ M3 m3_t(const M3& A) {
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
    return R;
}

__constant__ float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                             0.5462742152960396f};
__constant__ float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f

// Phase 1 of the kernel below: the sum of the partial-gradient rows of each of the workgroup's 256 Gaussians, into LDS.
// The backward blend (blend_bwd_q.hip) numbers its rows DENSELY in index order: wave w of this workgroup (the same 64
// consecutive Gaussians emit_kernel's wave w owned) has its rows from wave_rowbase[w] on, Gaussian by Gaussian
// (gauss_rows[] rows each), and every row is written.  So a wave STREAMS one contiguous range: 64 rows per window as RQ
// fully coalesced float4 loads per lane (lane l takes float4 l, l + 64, ... of the window), parked in LDS, then RQ lanes per
// Gaussian add the rows of their Gaussians -- no validity bytes, no holes, no per-instance gather.  Fixed summation order
// (the dense numbering: quadrants ascending within an instance, instances in emission order): bitwise reproducible.
template <int RQ>  // float4 per row (rowf / 4: 3, 4, 5 or 6)
struct ReduceLds {
    float4 s_row[4][GS2M_WAVE * RQ];  // one window per wave: 64 rows x RQ float4, row-major
    float4 s_sum[256 * RQ];           // the result: RQ float4 per Gaussian of the workgroup.  Until a Gaussian's sum is written, the first
                                      // two words of its slot hold its first row inside the wave's run and its row count -- whoever needs
                                      // them is working towards that very sum -- which keeps the block at 40 KB at RQ = 5: four workgroups per CU
};
template <int RQ, int WIN>  // WIN: windows in flight (2 or 3) -> the thread's own gauss_rows entry
__device__ __forceinline__ uint32_t reduce_rows_to_lds(int P, const uint32_t* __restrict__ gauss_rows, const uint32_t* __restrict__ wave_rowbase,
                                                       const float* __restrict__ rows, ReduceLds<RQ>& L) {
    constexpr int MAXQ = RQ, rq = RQ;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t cnt = 0;
    if (i < P) cnt = gauss_rows[i];
    const uint32_t own = cnt;
    // heavy Gaussians (common.h: GS2M_ROWS_BIG | first unit): their rows are not part of the wave's run; heavy_reduce_kernel below
    // has added them up per unit, the Gaussian's thread adds the units' sums (gaussian_bwd_kernel)
    if ((cnt & GS2M_ROWS_BIG) != 0u) cnt = 0;
    const uint32_t incl = wave_inclusive_scan_u32(cnt, lane);
    const uint32_t total = __shfl(incl, 63, 64);                      // rows of this wave's Gaussians
    const size_t wave_id = (size_t)blockIdx.x * 4 + wave;
    const uint32_t wb = wave_id * GS2M_WAVE < (size_t)P ? wave_rowbase[wave_id] : 0u;  // the wave's first row (binning.hip: rowscan_kernel)
    uint2* const s_run = reinterpret_cast<uint2*>(&L.s_sum[(wave * GS2M_WAVE) * rq]);  // Gaussian j of the wave: s_run[j * rq * 2] = {first row, rows}
    s_run[lane * rq * 2] = make_uint2(incl - cnt, cnt);
    const int G = GS2M_WAVE / rq, g = lane / rq, c = lane - g * rq;
    const bool worker = g < G;
    uint32_t j = worker ? (uint32_t)g : GS2M_WAVE;
    float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t nwin = (total + GS2M_WAVE - 1) / GS2M_WAVE;
    const float4* r4 = reinterpret_cast<const float4*>(rows) + (size_t)wb * rq;  // the wave's rows as one float4 stream
    const uint32_t nq = total * (uint32_t)rq;
    float4* const s_row = L.s_row[wave];
    auto load_window = [&](uint32_t w, float4* a) {
#pragma unroll
        for (int e = 0; e < MAXQ; e++) {
            const uint32_t q = w * (GS2M_WAVE * (uint32_t)rq) + (uint32_t)e * GS2M_WAVE + lane;
            a[e] = q < nq ? gs2m_ldnt(r4 + q) : make_float4(0.f, 0.f, 0.f, 0.f);  // read once: common.h, gs2m_ldnt
        }
    };
    // One window: park the loaded rows in LDS, then every group adds the rows of its Gaussians that lie in the window.
    // LDS operations of one wave execute in order: the reads see the writes above them.
    // A Gaussian that covers WHOLE windows (a splat over a few hundred tiles owns thousands of rows): its group alone would add
    // 64 rows per window from LDS, window after window, while the rest of the wave waits.  Such windows never go through LDS:
    // float4 number e * 64 + lane of every window belongs to channel quad (e * 64 + lane) mod rq whatever the window, so every
    // lane adds the windows into rq accumulators of its own, and when the run of covered windows ends the owner's lanes add
    // the 64 rq accumulators of their channel in index order -- a fixed order as well.
    float4 hacc[MAXQ];
#pragma unroll
    for (int e = 0; e < MAXQ; e++) hacc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    int heavy_owner = -1;  // group whose Gaussian the accumulators belong to (wave-uniform); -1: none
    auto consume = [&](uint32_t w, const float4* a) {
        const uint32_t k0 = w * GS2M_WAVE, k1 = w < nwin ? k0 + GS2M_WAVE : 0xFFFFFFFFu;
        bool mine = false;
        if (w < nwin && j < GS2M_WAVE) {
            const uint2 run = s_run[j * rq * 2];
            const uint32_t ex = run.x, cn = run.y;
            mine = ex <= k0 && ex + cn >= k1;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(mine && c == 0);
        const int cover = m != 0ull ? (__ffsll((long long)m) - 1) / rq : -1;  // group whose Gaussian owns the whole window (wave-uniform)
        if (heavy_owner >= 0 && cover != heavy_owner) {  // the run of covered windows has ended: hand the accumulators to their owner
#pragma unroll
            for (int e = 0; e < MAXQ; e++) {
                s_row[e * GS2M_WAVE + lane] = hacc[e];
                hacc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (worker && g == heavy_owner) {
                for (int q = c; q < GS2M_WAVE * rq; q += rq) {
                    const float4 v = s_row[q];
                    racc.x += v.x; racc.y += v.y; racc.z += v.z; racc.w += v.w;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            heavy_owner = -1;
        }
        if (cover >= 0) {  // registers only; the owner writes its sum once a later window takes the ordinary path
            heavy_owner = cover;
#pragma unroll
            for (int e = 0; e < MAXQ; e++) { hacc[e].x += a[e].x; hacc[e].y += a[e].y; hacc[e].z += a[e].z; hacc[e].w += a[e].w; }
            return;
        }
        if (w < nwin) {
#pragma unroll
            for (int e = 0; e < MAXQ; e++) s_row[e * GS2M_WAVE + lane] = a[e];
        }
        while (j < GS2M_WAVE) {
            const uint2 run = s_run[j * rq * 2];
            const uint32_t ex = run.x, cn = run.y;
            if (ex >= k1 && cn > 0) break;  // starts in a later window
            // rows of this Gaussian inside the window; those of windows it covered entirely are in racc already
            const uint32_t t0 = max(ex, k0), t1 = min(ex + cn, k1);
            for (uint32_t t = t0; t < t1; t++) {
                const float4 v = s_row[(t - k0) * rq + c];
                racc.x += v.x; racc.y += v.y; racc.z += v.z; racc.w += v.w;
            }
            if (ex + cn > k1) break;  // continues in the next window
            L.s_sum[(wave * GS2M_WAVE + (int)j) * rq + c] = racc;  // (a big Gaussian's entry: zeros here, its sum below)
            racc = make_float4(0.f, 0.f, 0.f, 0.f);
            j += (uint32_t)G;
        }
    };
    // WIN windows in flight: the windows of a wave are a serial chain (load -> LDS -> sum); with one window in flight every one of
    // them cost a full memory round trip.  Rounds 4-6 kept three: 139 registers and 43 + 49 KB of LDS = 3 waves per SIMD.  With TWO
    // windows -- 113 registers -- the run words inside the sum slots and dL/dSH leaving as basis x gradient, the kernel holds 4 waves per
    // SIMD: 154 -> 143.5 us at the bench size (the same kernel pinned to 3 waves: 168 us, to 2: 196).  A launch that does not fill the
    // chip anyway (the launcher: fewer than three workgroups per CU) keeps THREE: there the chain is all there is (10 k Gaussians:
    // 12.1 against 15.0 us).  One extra pass (w == nwin, an empty window) lets every group finish and write its remaining Gaussians.
    float4 a0[MAXQ], a1[MAXQ], a2[WIN > 2 ? MAXQ : 1];
    load_window(0, a0);
    load_window(1, a1);
    if (WIN > 2) load_window(2, a2);
    for (uint32_t w = 0; w <= nwin; w += WIN) {
        consume(w, a0);
        load_window(w + WIN, a0);
        if (w + 1 <= nwin) consume(w + 1, a1);
        load_window(w + WIN + 1, a1);
        if (WIN > 2) {
            if (w + 2 <= nwin) consume(w + 2, a2);
            load_window(w + WIN + 2, a2);
        }
    }
    return own;
}

// ---- heavy Gaussians: one wave per unit of 64 instances = 256 reserved rows, of which HeavyUnit::pop says which are written ----
// Lane l < NT takes float4 l, l + NT, ... of the unit's rows -- NT is a multiple of the row's float4 count, so a lane stays on one
// channel quad -- and the lanes' partials are added in lane order: a fixed order.  The unit's sum replaces its first row.
template <int RQ>
__global__ void __launch_bounds__(256) heavy_reduce_kernel(float* __restrict__ rows, const HeavyUnit* __restrict__ hrec, const uint32_t* __restrict__ counters) {
    constexpr int NT = (GS2M_WAVE / RQ) * RQ, RPL = NT / RQ /* rows per load */, NLOAD = (4 * (int)GS2M_UNIT + RPL - 1) / RPL;
    __shared__ float4 s_part[4][GS2M_WAVE];
    __shared__ uint8_t s_pop[4][GS2M_UNIT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t units = counters[GS2M_CNT_HUNITS];
    for (uint32_t u = blockIdx.x * 4u + (uint32_t)wave; u < units; u += gridDim.x * 4u) {
        s_pop[wave][lane] = hrec[u].pop[lane];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (one wave: LDS operations execute in order)
        float4* const r4 = reinterpret_cast<float4*>(rows) + (size_t)u * (4 * GS2M_UNIT) * RQ;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < NT) {
            const int c = lane % RQ, r0 = lane / RQ;
#pragma unroll 1
            for (int l0 = 0; l0 < NLOAD; l0 += 8) {
                float4 v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int r = (l0 + e) * RPL + r0;  // row of the unit: instance r >> 2, its (r & 3)-th row
                    const bool ok = l0 + e < NLOAD && r < 4 * (int)GS2M_UNIT && (uint32_t)(r & 3) < (uint32_t)s_pop[wave][r >> 2];
                    v[e] = ok ? r4[r * RQ + c] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int e = 0; e < 8; e++) { acc.x += v[e].x; acc.y += v[e].y; acc.z += v[e].z; acc.w += v[e].w; }
            }
        }
        s_part[wave][lane] = acc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < RQ) {  // channel quad `lane`: the partials of lanes lane, lane + RQ, ... in that order
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = lane; k < NT; k += RQ) {
                const float4 v = s_part[wave][k];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            r4[lane] = t;  // (every load of this wave has returned: the partials depend on them)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// dL/dSH on its way out: every thread of the workgroup has left basis[16] and dL/dRGB[3] of its Gaussian in LDS (row stride
// SHO_STRIDE = 19 floats: odd, conflict-free); element (k, c) of a Gaussian's (16,3) gradient is basis[k] * dL/dRGB[c].  The block's
// 256 rows are one contiguous run of the output tensor(s): written as fully coalesced non-temporal float4 stores, the products formed
// here.  Coefficients at or above `nact` (= (D + 1)^2) are +0.  Packed form (P,16,3) or the model's split form (P,1,3) + (P,15,3).
#define SHO_STRIDE 19
__device__ __forceinline__ float sh_outer_val(const float* __restrict__ s_bg, int row, int cc /* 0..47: 3 k + c */, int nact) {
    const int k = cc / 3, c = cc - 3 * k;
    const float* r = s_bg + row * SHO_STRIDE;
    return k < nact ? r[k] * r[16 + c] : 0.f;
}
__device__ __forceinline__ void sh_outer_store(float* __restrict__ dshs, float* __restrict__ drest, int P, const float* __restrict__ s_bg, int nact) {
    const int tid = threadIdx.x;
    if (drest == nullptr) {
        const size_t base4 = (size_t)blockIdx.x * 256 * 12, lim4 = (size_t)P * 12;
        float4* o4 = reinterpret_cast<float4*>(dshs);
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const size_t k = base4 + tid + 256 * i;
            const int e = 4 * (tid + 256 * i);
            const int row = e / 48, col = e - row * 48;
            if (k < lim4) gs2m_stnt(o4 + k, make_float4(sh_outer_val(s_bg, row, col, nact), sh_outer_val(s_bg, row, col + 1, nact),
                                                        sh_outer_val(s_bg, row, col + 2, nact), sh_outer_val(s_bg, row, col + 3, nact)));
        }
    } else {
        const size_t dbase = (size_t)blockIdx.x * 768, dlim = (size_t)P * 3;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int e = tid + 256 * i;
            if (dbase + e < dlim) dshs[dbase + e] = sh_outer_val(s_bg, e / 3, e % 3, nact);
        }
        // rest: float4 k4 = tid + 256 i of the block's 2880; element e = 4 k4 + j sits in row e / 45 at coefficient 3 + e % 45
        const size_t rbase = (size_t)blockIdx.x * 11520, rlim = (size_t)P * 45;
        const bool full = rbase + 11520 <= rlim;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int k4 = tid + 256 * i;
            if (k4 < 2880) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int e = 4 * k4 + j, row = e / 45;
                    v[j] = sh_outer_val(s_bg, row, 3 + (e - 45 * row), nact);
                }
                const size_t ge = rbase + 4 * (size_t)k4;
                if (full || ge + 3 < rlim) {
                    gs2m_stnt(reinterpret_cast<float4*>(drest + ge), make_float4(v[0], v[1], v[2], v[3]));
                } else {
                    if (ge < rlim) drest[ge] = v[0];
                    if (ge + 1 < rlim) drest[ge + 1] = v[1];
                    if (ge + 2 < rlim) drest[ge + 2] = v[2];
                }
            }
        }
    }
}

template <bool SH_LDS, int RQ, int WIN>
__global__ void __launch_bounds__(256) gaussian_bwd_kernel(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ shs_rest, const float* __restrict__ colors_precomp, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, const float* __restrict__ vm,
    const float* __restrict__ proj, const float* __restrict__ campos, float h_x, float h_y, float tan_fovx,
    float tan_fovy, const int* __restrict__ radii, int fc, const float4* __restrict__ rec,
    const uint32_t* __restrict__ gauss_rows, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ wave_rowbase,
    const uint8_t* __restrict__ clamped,
    const float* __restrict__ sh_dir, const float* __restrict__ rows /* nullptr: nothing was rendered, every sum is zero */, int want_sh,
    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dconics,
    float* __restrict__ dL_dopacities, float* __restrict__ dL_dcolors, float* __restrict__ dL_dmeans3D,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dshs, float* __restrict__ dL_dshs_rest, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drots, float* __restrict__ dL_dfeatures) {
    // dL/dSH goes out through LDS so that every global access of the (P,16,3) tensor is a coalesced stream: per Gaussian the 16 basis
    // values and dL/dRGB (row stride SHO_STRIDE = 19 floats), multiplied out by the store pass (sh_outer_store).  The coefficients
    // themselves are not read here: all the backward needs of them is d(colour)/d(direction), 9 floats per Gaussian that the
    // forward's preprocess kernel left in GeomState::sh_dir.
    // one LDS block, two lives: the row sums (phase 1), then the block's dL/dSH factors on their way out
    constexpr size_t kShBytes = SH_LDS ? 256 * SHO_STRIDE * sizeof(float) : 16, kRedBytes = sizeof(ReduceLds<RQ>);
    __shared__ __align__(16) unsigned char s_raw[kShBytes > kRedBytes ? kShBytes : kRedBytes];
    float* const s_sh = reinterpret_cast<float*>(s_raw);
    ReduceLds<RQ>& red = *reinterpret_cast<ReduceLds<RQ>*>(s_raw);
    float acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0.f;
    if (rows != nullptr) {
        const uint32_t gr = reduce_rows_to_lds<RQ, WIN>(P, gauss_rows, wave_rowbase, rows, red);
        gs2m_sync();
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            const float4 v = red.s_sum[threadIdx.x * RQ + q];
            acc[4 * q] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
        }
        gs2m_sync();  // the sums are in registers: the LDS block may be overwritten
        // heavy Gaussians: the sums of their units (heavy_reduce_kernel).  Up to four units: the Gaussian's own thread adds them, in unit
        // order (a crowded wave holds dozens of one-unit Gaussians: all of them at once).  More (a splat over hundreds of tiles): the
        // wave's 64 lanes fetch them together, lane l units l, l + 64, ..., and a butterfly adds the lanes up -- a fixed order as well.
        const int hi = blockIdx.x * 256 + threadIdx.x;
        const bool hv = (gr & GS2M_ROWS_BIG) != 0u;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hv) != 0ull, 0)) {  // (wave-uniform; no heavy Gaussian in most waves)
            const uint32_t u0 = gr & ~GS2M_ROWS_BIG, nu = hv ? (tiles_touched[hi] + GS2M_UNIT - 1u) / GS2M_UNIT : 0u;
            const float4* r4 = reinterpret_cast<const float4*>(rows);
            if (hv && nu <= 4u) {
                for (uint32_t j = 0; j < nu; j++) {
#pragma unroll
                    for (int q = 0; q < RQ; q++) {
                        const float4 v = r4[(size_t)(u0 + j) * (4 * GS2M_UNIT) * RQ + q];
                        acc[4 * q] += v.x; acc[4 * q + 1] += v.y; acc[4 * q + 2] += v.z; acc[4 * q + 3] += v.w;
                    }
                }
            }
            unsigned long long many = __builtin_amdgcn_ballot_w64(hv && nu > 4u);
            const int lane = threadIdx.x & 63;
            while (many != 0ull) {
                const int L = __builtin_ctzll(many);
                many &= many - 1ull;
                const uint32_t u0L = (uint32_t)__shfl((int)u0, L, 64), nuL = (uint32_t)__shfl((int)nu, L, 64);
                float part[4 * RQ];
#pragma unroll
                for (int k = 0; k < 4 * RQ; k++) part[k] = 0.f;
                for (uint32_t j = (uint32_t)lane; j < nuL; j += GS2M_WAVE) {
#pragma unroll
                    for (int q = 0; q < RQ; q++) {
                        const float4 v = r4[(size_t)(u0L + j) * (4 * GS2M_UNIT) * RQ + q];
                        part[4 * q] += v.x; part[4 * q + 1] += v.y; part[4 * q + 2] += v.z; part[4 * q + 3] += v.w;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4 * RQ; k++) {
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) part[k] += __shfl_xor(part[k], d, 64);
                }
                if (lane == L) {
#pragma unroll
                    for (int k = 0; k < 4 * RQ; k++) acc[k] += part[k];
                }
            }
        }
    }
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = idx < P;
    const int li = in_range ? idx : 0;
    const bool visible = in_range && radii[li] > 0;
    const float mx = means3D[3 * li], my = means3D[3 * li + 1], mz = means3D[3 * li + 2];
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    float s_in[3] = {0.f, 0.f, 0.f};
    if (scales != nullptr) {
        q_in = reinterpret_cast<const float4*>(rotations)[li];
        s_in[0] = scales[3 * li]; s_in[1] = scales[3 * li + 1]; s_in[2] = scales[3 * li + 2];
    }
    float sdv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (visible && shs != nullptr && D > 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) sdv[k] = sh_dir[9 * (size_t)li + k];
    }
    if (!SH_LDS && !in_range) return;
    if (in_range) {

    // ---- gradients that are plain sums of the rows ----
    reinterpret_cast<float4*>(dL_dmeans2D)[idx] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (dL_dconics) reinterpret_cast<float4*>(dL_dconics)[idx] = make_float4(acc[4], acc[5], 0.f, acc[6]);
    dL_dopacities[idx] = acc[7];
    if (dL_dcolors) {  // wanted only when the colours came in precomputed (the SH chain below uses the row itself)
        dL_dcolors[3 * idx] = acc[ROW_COL]; dL_dcolors[3 * idx + 1] = acc[ROW_COL + 1]; dL_dcolors[3 * idx + 2] = acc[ROW_COL + 2];
    }
#pragma unroll
    for (int ch = 0; ch < GS2M_NUM_FEATURES; ch++)
        dL_dfeatures[(size_t)idx * GS2M_NUM_FEATURES + ch] = ch < fc ? acc[ROW_FEAT + ch] : 0.f;

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};

    if (visible) {
        // ---- 3D covariance (recomputed exactly as preprocess.hip does) ----
        float c3[6];
        float sx = 0.f, sy = 0.f, sz = 0.f, qr = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
        M3 R, Mm;
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * (size_t)idx + k];
        } else {
            sx = scale_modifier * s_in[0]; sy = scale_modifier * s_in[1]; sz = scale_modifier * s_in[2];
            qr = q_in.x; qx = q_in.y; qy = q_in.z; qz = q_in.w;
            const float r = qr, x = qx, y = qy, z = qz;
            M3 S = m3_cols(sx, 0.f, 0.f, 0.f, sy, 0.f, 0.f, 0.f, sz);
            R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            Mm = m3_mul(S, R);
            M3 Sig = m3_mul(m3_t(Mm), Mm);
            c3[0] = Sig.m[0][0]; c3[1] = Sig.m[0][1]; c3[2] = Sig.m[0][2];
            c3[3] = Sig.m[1][1]; c3[4] = Sig.m[1][2]; c3[5] = Sig.m[2][2];
        }
        // ---- computeCov2DCUDA ----
        const float dcx = acc[4], dcy = acc[5], dcz = acc[6];
        float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
        float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
        const float tz_ = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float txtz = tx / tz_, tytz = ty / tz_;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz_;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz_;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        M3 J = m3_cols(h_x / tz_, 0.0f, -(h_x * tx) / (tz_ * tz_), 0.0f, h_y / tz_, -(h_y * ty) / (tz_ * tz_), 0.f, 0.f, 0.f);
        M3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
        M3 Vrk = m3_cols(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
        M3 T = m3_mul(Wm, J);
        M3 cov2D = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
        const float a = cov2D.m[0][0] + 0.3f;  // low-pass appears only in the backward (backward.cu:205-207)
        const float b = cov2D.m[0][1];
        const float c = cov2D.m[1][1] + 0.3f;
        const float denom = a * c - b * b;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) T.m[i][j]
#define VV(i, j) Vrk.m[i][j]
#define WW(i, j) Wm.m[i][j]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
            dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
            dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
            dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
            dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
            dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
            dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
        }
        const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                              (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
        const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                              (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
        const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                              (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
        const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                              (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
        const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                              (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
        const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                              (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
        const float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
        const float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
        const float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
        const float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
        const float tz = 1.f / tz_;
        const float tz2 = tz * tz;
        const float tz3 = tz2 * tz;
        const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 + (2 * h_y * ty) * tz3 * dL_dJ12;
        dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

        // ---- projection of the 2D mean (backward.cu:376-392) ----
        {
            const float m_hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
            const float m_w = 1.0f / (m_hw + 0.0000001f);
            const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
            const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
            const float gx = acc[0], gy = acc[1];
            dmean[0] += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
            dmean[1] += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
            dmean[2] += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        }

        // ---- SH colour backward (backward.cu:23-148) ----
        if (shs != nullptr) {
            const float ox = mx - campos[0], oy = my - campos[1], oz = mz - campos[2];
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            const float x = ox / len, y = oy / len, z = oz / len;
            float* dsh = SH_LDS ? (s_sh + threadIdx.x * SHO_STRIDE) : (dL_dshs + (size_t)idx * M * 3);
            const uint8_t cl = clamped[idx];
            float g[3] = {acc[ROW_COL] * ((cl & 1) ? 0.f : 1.f), acc[ROW_COL + 1] * ((cl & 2) ? 0.f : 1.f),
                          acc[ROW_COL + 2] * ((cl & 4) ? 0.f : 1.f)};
            // d(colour)/d(direction), evaluated by the forward (preprocess.hip) with the reference's expressions (backward.cu:62-146)
            const float dRdx[3] = {sdv[0], sdv[1], sdv[2]}, dRdy[3] = {sdv[3], sdv[4], sdv[5]}, dRdz[3] = {sdv[6], sdv[7], sdv[8]};
            // dL/dSH_k = basis_k(direction) x dL/dRGB: with the LDS path a thread leaves its 16 basis values and the three gradient
            // components (19 floats instead of the 48 products: 19 KB per workgroup instead of 49) and the products are formed by the
            // coalesced store pass (sh_outer_store below) -- the same fp32 multiplications
#define DSH(k, val)                                           \
    do {                                                      \
        const float v_ = (val);                               \
        if (SH_LDS) {                                         \
            dsh[(k)] = v_;                                    \
        } else {                                              \
            dsh[(k) * 3 + 0] = v_ * g[0];                     \
            dsh[(k) * 3 + 1] = v_ * g[1];                     \
            dsh[(k) * 3 + 2] = v_ * g[2];                     \
        }                                                     \
    } while (0)
            float xx = 0.f, yy = 0.f, zz = 0.f, xy = 0.f, yz = 0.f, xz = 0.f;
            if (D > 1) { xx = x * x; yy = y * y; zz = z * z; xy = x * y; yz = y * z; xz = x * z; }
            // pass 2: dL/dSH_k = basis_k(dir) * dL/dRGB (skipped when the caller did not ask for dL/dSH: include/gs2m_raster.h)
            if (want_sh) {
            DSH(0, SH_C0);
            if (D > 0) {
                DSH(1, -SH_C1 * y);
                DSH(2, SH_C1 * z);
                DSH(3, -SH_C1 * x);
                if (D > 1) {
                    DSH(4, kC2[0] * xy);
                    DSH(5, kC2[1] * yz);
                    DSH(6, kC2[2] * (2.f * zz - xx - yy));
                    DSH(7, kC2[3] * xz);
                    DSH(8, kC2[4] * (xx - yy));
                    if (D > 2) {
                        DSH(9, kC3[0] * y * (3.f * xx - yy));
                        DSH(10, kC3[1] * xy * z);
                        DSH(11, kC3[2] * y * (4.f * zz - xx - yy));
                        DSH(12, kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                        DSH(13, kC3[4] * x * (4.f * zz - xx - yy));
                        DSH(14, kC3[5] * z * (xx - yy));
                        DSH(15, kC3[6] * x * (xx - 3.f * yy));
                    }
                }
            }
            // coefficients above the active degree receive no gradient
            if (SH_LDS) {
                for (int k = (D + 1) * (D + 1); k < M; k++) dsh[k] = 0.f;  // (sh_outer_store writes +0 for them whatever the gradient's sign)
                dsh[16] = g[0]; dsh[17] = g[1]; dsh[18] = g[2];
            } else {
                for (int k = (D + 1) * (D + 1); k < M; k++) { dsh[k * 3] = 0.f; dsh[k * 3 + 1] = 0.f; dsh[k * 3 + 2] = 0.f; }
            }
            }
#undef DSH
            const float ddx = dRdx[0] * g[0] + dRdx[1] * g[1] + dRdx[2] * g[2];
            const float ddy = dRdy[0] * g[0] + dRdy[1] * g[1] + dRdy[2] * g[2];
            const float ddz = dRdz[0] * g[0] + dRdz[1] * g[1] + dRdz[2] * g[2];
            // dnormvdv (auxiliary.h:111-120)
            const float sum2 = ox * ox + oy * oy + oz * oz;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((+sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
            dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
            dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
        }

        // ---- scale / rotation backward (backward.cu:285-347) ----
        if (scales != nullptr) {
            const float r = qr, x = qx, y = qy, z = qz;
            M3 dSig = m3_cols(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                              0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
            M3 M2;
#pragma unroll
            for (int c_ = 0; c_ < 3; c_++)
#pragma unroll
                for (int r_ = 0; r_ < 3; r_++) M2.m[c_][r_] = Mm.m[c_][r_] * 2.0f;
            M3 dM = m3_mul(M2, dSig);
            M3 Rt = m3_t(R);
            M3 dMt = m3_t(dM);
            dscale[0] = Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2];
            dscale[1] = Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2];
            dscale[2] = Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2];
#pragma unroll
            for (int j = 0; j < 3; j++) { dMt.m[0][j] *= sx; dMt.m[1][j] *= sy; dMt.m[2][j] *= sz; }
#define Dm(i, j) dMt.m[i][j]
            drot[0] = 2 * z * (Dm(0, 1) - Dm(1, 0)) + 2 * y * (Dm(2, 0) - Dm(0, 2)) + 2 * x * (Dm(1, 2) - Dm(2, 1));
            drot[1] = 2 * y * (Dm(1, 0) + Dm(0, 1)) + 2 * z * (Dm(2, 0) + Dm(0, 2)) + 2 * r * (Dm(1, 2) - Dm(2, 1)) - 4 * x * (Dm(2, 2) + Dm(1, 1));
            drot[2] = 2 * x * (Dm(1, 0) + Dm(0, 1)) + 2 * r * (Dm(2, 0) - Dm(0, 2)) + 2 * z * (Dm(1, 2) + Dm(2, 1)) - 4 * y * (Dm(2, 2) + Dm(0, 0));
            drot[3] = 2 * r * (Dm(0, 1) - Dm(1, 0)) + 2 * x * (Dm(2, 0) + Dm(0, 2)) + 2 * y * (Dm(1, 2) + Dm(2, 1)) - 4 * z * (Dm(1, 1) + Dm(0, 0));
#undef Dm
        }
    } else if (shs != nullptr && M > 0 && want_sh) {
        if (SH_LDS) {
            float* dsh = s_sh + threadIdx.x * SHO_STRIDE;
            for (int k = 0; k < SHO_STRIDE; k++) dsh[k] = 0.f;
        } else {
            float* dsh = dL_dshs + (size_t)idx * M * 3;
            for (int k = 0; k < 3 * M; k++) dsh[k] = 0.f;
        }
    }

    dL_dmeans3D[3 * idx] = dmean[0]; dL_dmeans3D[3 * idx + 1] = dmean[1]; dL_dmeans3D[3 * idx + 2] = dmean[2];
    if (dL_dcov3D) {  // wanted only when the covariances came in precomputed
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)idx + k] = dcov[k];
    }
    dL_dscales[3 * idx] = dscale[0]; dL_dscales[3 * idx + 1] = dscale[1]; dL_dscales[3 * idx + 2] = dscale[2];
    reinterpret_cast<float4*>(dL_drots)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }  // in_range
    if (SH_LDS) {
        gs2m_sync();  // every thread has replaced its LDS row by its dL/dSH row
        sh_outer_store(dL_dshs, dL_dshs_rest, P, s_sh, (D + 1) * (D + 1));
    }
}

}  // namespace

void gs2m_launch_gaussian_bwd(int P, int D, int M, const float* means3D, const float* shs, const float* shs_rest,
                              const float* colors_precomp,
                              const float* scales, float scale_modifier, const float* rotations,
                              const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                              const float* campos, int W, int H, float tan_fovx, float tan_fovy, const int* radii,
                              int fc, const GeomState& g, const float* rows, int rowf, bool have_rows,
                              float* dL_dmeans2D, float* dL_dconics, float* dL_dopacities, float* dL_dcolors,
                              float* dL_dmeans3D, float* dL_dcov3D, float* dL_dshs, float* dL_dshs_rest, float* dL_dscales,
                              float* dL_drots, float* dL_dfeatures, hipStream_t s) {
    const float h_x = W / (2.0f * tan_fovx), h_y = H / (2.0f * tan_fovy);
#define GS2M_GB(LDS, RQ, WIN)                                                                                           \
    gaussian_bwd_kernel<LDS, RQ, WIN><<<(P + 255) / 256, 256, 0, s>>>(                                                  \
        P, D, M, means3D, shs, shs_rest, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,   \
        projmatrix, campos, h_x, h_y, tan_fovx, tan_fovy, radii, fc, g.rec, g.gauss_rows, g.tiles_touched, g.wave_rowbase, g.clamped, \
        g.sh_dir, have_rows ? rows : nullptr, want_sh, dL_dmeans2D, dL_dconics, dL_dopacities, dL_dcolors, dL_dmeans3D, dL_dcov3D, \
        dL_dshs, dL_dshs_rest, dL_dscales, dL_drots, dL_dfeatures)
#define GS2M_GBW(LDS, RQ)                                                                                               \
    if (few) GS2M_GB(LDS, RQ, 3); else GS2M_GB(LDS, RQ, 2)
#define GS2M_GBQ(LDS)                                                                                                   \
    switch (rowf >> 2) {                                                                                                \
        case 3: GS2M_GBW(LDS, 3); break;                                                                                \
        case 4: GS2M_GBW(LDS, 4); break;                                                                                \
        case 5: GS2M_GBW(LDS, 5); break;                                                                                \
        default: GS2M_GBW(LDS, 6); break;                                                                               \
    }
    const bool few = (P + 255) / 256 < 3 * 256;  // fewer workgroups than three per CU: occupancy is not the limit, the wave's chain is
    // dL_dshs == NULL with SH input: the caller does not want dL/dSH (its colour gradient is identically zero, e.g. a view rendered for
    // its depth and normals only): the per-Gaussian kernel skips the 48 stores per Gaussian; everything else is computed as usual
    const int want_sh = dL_dshs != nullptr ? 1 : 0;
    const bool lds = want_sh && shs != nullptr && M == 16 &&
                     (shs_rest ? ((((uintptr_t)shs_rest) | ((uintptr_t)dL_dshs_rest)) & 15) == 0
                               : ((((uintptr_t)shs) | ((uintptr_t)dL_dshs)) & 15) == 0);
    if (lds) { GS2M_GBQ(true) } else { GS2M_GBQ(false) }
#undef GS2M_GBQ
#undef GS2M_GBW
#undef GS2M_GB
}

void gs2m_launch_heavy_reduce(float* rows, int rowf, const BinningState& b, const GeomState& g, long long heavy_units, hipStream_t s) {
    if (heavy_units == 0 || rows == nullptr) return;
    // known on the host: a wave per unit; otherwise a fixed grid that reads the count on the device (and leaves at once without units)
    const unsigned grid = heavy_units > 0 ? (unsigned)((heavy_units + 3) / 4 < 8192 ? (heavy_units + 3) / 4 : 8192) : 512u;
    switch (rowf >> 2) {
        case 3: heavy_reduce_kernel<3><<<grid, 256, 0, s>>>(rows, b.hrec, g.counters); break;
        case 4: heavy_reduce_kernel<4><<<grid, 256, 0, s>>>(rows, b.hrec, g.counters); break;
        case 5: heavy_reduce_kernel<5><<<grid, 256, 0, s>>>(rows, b.hrec, g.counters); break;
        default: heavy_reduce_kernel<6><<<grid, 256, 0, s>>>(rows, b.hrec, g.counters); break;
    }
}
