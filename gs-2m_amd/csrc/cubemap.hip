// Environment-map prefilters of the deferred PBR stage for gfx950 (the ROCm replacement of
// `render_utils.diffuse_cubemap` / `specular_cubemap`, called by CubemapLight.build_mips for every training view,
// pbr/light.py:86-99; kernels in submodules/render-utils/c_src/cubemap.cu:110-350).
//
//   diffuse   out[o] = sum over ALL texels t of  clamp(N_o . L_t, 0, 0.999) * area(t) / 3.141592 * in[t]
//   specular  out[o] = sum over texels t with L_t . V_o >= cutoff of  w(o, t) * in[t],  4th channel sum of w,
//             w(o, t) = max(L_t . V_o, 0) * D_ggx(roughness^4; max(V_o . H, 0)) * area(t) / 4,  H = normalize(L_t + V_o)
// with L_t / N_o / V_o the normalised texel-centre directions (face table of pbr/light.py:13-26) and area(t) the
// texel's solid-angle proxy atan((x+1)/h) - atan(x/h) per axis (cubemap.cu:17-30).
//
// Both backward passes are gathers as well, not scatters: the pair geometry is symmetric (L_t . V_o, H), only area(t)
// belongs to one side, so   grad_in[t] = area(t) * sum over o of k(o, t) * grad_out[o]   with the same cone test --
// no atomics, deterministic.  The reference accelerates the specular sum with a table of per-(texel, face) bounding
// boxes found by brute force once per resolution; the boxes do not change the result (every texel inside is still
// tested against the cutoff), so here the box of the cone on each face is computed analytically per thread: the
// face coordinate a = x/z is the tangent of the longitude about the face's y axis, and a cone of half-angle theta about
// V spans longitudes psi_V +- asin(sin(theta) / |V_xz|).
//
// 1, 16 or 64 lanes per output texel by the size of the cone (a 512^2 level with a sub-texel lobe has 1.5M outputs of
// ~11 pairs, a 128^2 level with a 15-degree lobe 98k outputs of ~1600).  A pair costs ~5 operations for the cone test
// (directions are recomputed, not loaded: loading a 16-byte table row per pair made the kernel L2-bandwidth bound)
// and, inside the cone, a 12-byte colour load and ~12 more operations.
#include "common.h"
#include "../../include/gs2m_cubemap.h"

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 safe_normalize(V3 v) {  // render-utils c_src/vec3f.h:90-94
    const float l = sqrtf(dot3(v, v));
    return l > 0.f ? v3(v.x / l, v.y / l, v.z / l) : v3(0.f, 0.f, 0.f);
}

__device__ __forceinline__ V3 face_point(int side, float fx, float fy) {  // unnormalised texel direction
    switch (side) {
        case 0: return v3(1.f, -fy, -fx);
        case 1: return v3(-1.f, -fy, fx);
        case 2: return v3(fx, 1.f, fy);
        case 3: return v3(fx, -1.f, -fy);
        case 4: return v3(fx, -fy, 1.f);
        default: return v3(-fx, -fy, -1.f);
    }
}

__device__ __forceinline__ V3 texel_dir(int x, int y, int side, int N) {  // cubemap.cu:32-46
    const float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
    const float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
    switch (side) {
        case 0: return safe_normalize(v3(1.f, -fy, -fx));
        case 1: return safe_normalize(v3(-1.f, -fy, fx));
        case 2: return safe_normalize(v3(fx, 1.f, fy));
        case 3: return safe_normalize(v3(fx, -1.f, -fy));
        case 4: return safe_normalize(v3(fx, -fy, 1.f));
        default: return safe_normalize(v3(-fx, -fy, -1.f));
    }
}

// world direction -> the face's own frame, in which the face is the plane z = 1 and (x, y) are the face coordinates
__device__ __forceinline__ V3 to_face_frame(int side, V3 v) {
    switch (side) {
        case 0: return v3(-v.z, -v.y, v.x);
        case 1: return v3(v.z, -v.y, -v.x);
        case 2: return v3(v.x, v.z, v.y);
        case 3: return v3(v.x, -v.z, -v.y);
        case 4: return v3(v.x, -v.y, v.z);
        default: return v3(-v.x, -v.y, -v.z);
    }
}

__device__ __forceinline__ float texel_area(int x, int y, int N) {  // cubemap.cu:17-30
    if (N <= 1) return 1.f;
    const int H = N / 2;
    x = abs(x - H);
    y = abs(y - H);
    const float dx = atanf((float)(x + 1) / (float)H) - atanf((float)x / (float)H);
    const float dy = atanf((float)(y + 1) / (float)H) - atanf((float)y / (float)H);
    return dx * dy;
}

// texel index range [lo, hi] along one face axis that can contain directions within the cone (sin_t = sin of the half
// angle) about c = V in the face frame; `u` is the in-plane component of that axis.  Empty: lo > hi.
__device__ __forceinline__ void cone_range(float u, float cz, float sin_t, int N, int& lo, int& hi) {
    const float rho = sqrtf(u * u + cz * cz);
    if (sin_t >= rho * 0.999f) { lo = 0; hi = N - 1; return; }  // the cone contains the axis' pole: every longitude
    const float psi = atan2f(u, cz), delta = asinf(sin_t / rho) + 1e-4f;
    const float lim = 1.5607963f;  // pi/2 - 0.01: the face itself only spans |longitude| <= pi/4
    float a0 = psi - delta, a1 = psi + delta;
    if (a1 < -lim || a0 > lim) { lo = 1; hi = 0; return; }
    a0 = tanf(fmaxf(a0, -lim));
    a1 = tanf(fminf(a1, lim));
    if (a0 > 1.f || a1 < -1.f) { lo = 1; hi = 0; return; }
    lo = max((int)floorf((fmaxf(a0, -1.f) + 1.f) * (0.5f * (float)N)) - 1, 0);
    hi = min((int)floorf((fminf(a1, 1.f) + 1.f) * (0.5f * (float)N)) + 1, N - 1);
}

// ---------------------------------------------------------------- diffuse
template <bool BWD>
__global__ void __launch_bounds__(64) diffuse_kernel(int N, const float* __restrict__ in, float* __restrict__ out) {
    // one wave per output texel, lanes stride over the 6 N^2 inputs
    const int o = blockIdx.x;
    const int os = o / (N * N), oy = (o / N) % N, ox = o % N;
    const V3 No = texel_dir(ox, oy, os, N);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int total = 6 * N * N;
    for (int t = threadIdx.x; t < total; t += 64) {
        const int s = t / (N * N), y = (t / N) % N, x = t % N;
        const V3 L = texel_dir(x, y, s, N);
        const float c = fminf(fmaxf(dot3(No, L), 0.f), 0.999f);
        // forward: weight of input t in output o carries area(t); backward (o is the INPUT texel now): area(o), below
        const float w = BWD ? c : c * texel_area(x, y, N) / 3.141592f;
        a0 += w * in[3 * t]; a1 += w * in[3 * t + 1]; a2 += w * in[3 * t + 2];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a0 += __shfl_xor(a0, d); a1 += __shfl_xor(a1, d); a2 += __shfl_xor(a2, d);
    }
    if (threadIdx.x == 0) {
        const float k = BWD ? texel_area(ox, oy, N) / 3.141592f : 1.f;
        out[3 * o] = a0 * k; out[3 * o + 1] = a1 * k; out[3 * o + 2] = a2 * k;
    }
}

// ---------------------------------------------------------------- specular
// The texel directions need no table: in the frame of face s (face = plane z = 1) the texel centre is (fx, fy, 1) and
//   L . V = (P_o . P_t) / (|P_o| |P_t|),   P = (+-1, +-fx, +-fy) permuted by the face,
// three multiplies, two adds, a reciprocal square root and two multiplies per pair.  The area proxy is separable,
// area(x, y) = ax[x] * ax[y]; ax is the one table (res floats), and for unit vectors V . H = sqrt((1 + L . V) / 2).
__global__ void __launch_bounds__(256) axis_area_kernel(int N, float* __restrict__ ax) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= N) return;
    if (N <= 1) { ax[x] = 1.f; return; }
    const int H = N / 2, k = abs(x - H);
    ax[x] = atanf((float)(k + 1) / (float)H) - atanf((float)k / (float)H);
}

// forward: o = output texel, gathers inputs (3 channels in, 4 out); backward: o = input texel, gathers output gradients
// (4 channels in, of which the 4th -- d/d wsum -- does not reach the cubemap; 3 out).  TPO lanes share one output texel
// and stride over the box of every face (wide lobes on small levels would otherwise leave the chip empty).
template <bool BWD, int TPO>
__global__ void __launch_bounds__(256) specular_kernel(int N, float roughness, float cos_cut, const float* __restrict__ ax,
                                                       const float* __restrict__ in, float* __restrict__ out) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 6 * N * N;
    const int o_raw = gt / TPO, sub = gt % TPO;
    const bool live = o_raw < total;
    const int o = live ? o_raw : total - 1;  // keep whole groups converged for the reduction
    const int os = o / (N * N), oy = (o / N) % N, ox = o % N;
    const V3 Vo = texel_dir(ox, oy, os, N);
    const float alpha = roughness * roughness, alpha_sqr = alpha * alpha;
    const float sin_t = sqrtf(fmaxf(1.f - cos_cut * cos_cut, 0.f));
    constexpr int IC = BWD ? 4 : 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ws = 0.f;
    // D_ggx at cos^2(theta_h) = V.H^2 = (1 + L.V) / 2 for unit vectors: no half vector, no square root; the clamp of
    // ndfGGX (cubemap.cu:193-198) cannot bind for L.V in [cutoff, 1]
    const float pi_inv_a2 = alpha_sqr / 3.14159265358979323846f, a2m1h = 0.5f * (alpha_sqr - 1.f);
    // (the file is compiled with -ffp-contract=off: the FMAs are spelled out where rounding symmetry is not at stake)
    // (a branch-free form -- weight 0 outside the cone -- was measured: 3-13 % slower; the exec-mask regions do skip work)
    auto accumulate = [&](float d, float area, const float c0, const float c1, const float c2) {  // area: already x 1/4
        if (d >= cos_cut) {
            const float den = __builtin_fmaf(1.f + d, a2m1h, 1.f);          // cos^2 (alpha^2 - 1) + 1
            const float k = fmaxf(d, 0.f) * pi_inv_a2 * __builtin_amdgcn_rcpf(den * den);
            const float w = BWD ? k : k * area;
            a0 = __builtin_fmaf(w, c0, a0); a1 = __builtin_fmaf(w, c1, a1); a2 = __builtin_fmaf(w, c2, a2);
            ws += w;
        }
    };
    const float step = 2.0f / (float)N, org = 1.0f / (float)N - 1.0f;  // fx = org + step * x
    // L . V must come out bit-identical whichever of the two texels plays the output (the backward applies the same
    // cone test with the roles swapped; a pair accepted one way and rejected the other breaks the adjoint): products of
    // the UNNORMALISED face points summed in world-axis order, times the product of the two reciprocal lengths
    const float fxo = org + step * (float)ox, fyo = org + step * (float)oy;
    const V3 Po = face_point(os, fxo, fyo);
    const float ro = __builtin_amdgcn_rsqf((fxo * fxo + fyo * fyo) + 1.f);
    for (int s = 0; s < 6; s++) {
        const V3 c = to_face_frame(s, Vo);
        int x0, x1, y0, y1;
        if (cos_cut <= 0.70710678f) { x0 = 0; x1 = N - 1; y0 = 0; y1 = N - 1; }  // half-angle >= 45 deg: no useful box
        else {
            if (c.z <= -sin_t) continue;  // the whole cone is behind the face's plane
            // The cone misses the face's wedge |longitude| <= pi/4 along an axis when sqrt(2) rho sin(psi - pi/4) =
            // |u| - cz exceeds sqrt(2) sin_t (psi, rho: cone_range's polar coordinates).  cone_range finds the same by way
            // of atan2 / asin / tan -- ~300 instructions per axis, and with a narrow lobe (the 256^2 level: 28 steps of
            // box per output) the five faces it ends up rejecting cost twice what the box of the sixth costs.  The margin
            // (2e-3 against cone_range's 1e-4 rad) makes this a strict subset of its rejections: the boxes that remain,
            // and with them the order of every sum, are unchanged.
            if (fmaxf(fabsf(c.x), fabsf(c.y)) - c.z > 1.41421356f * sin_t + 2e-3f) continue;
            cone_range(c.x, c.z, sin_t, N, x0, x1);
            cone_range(c.y, c.z, sin_t, N, y0, y1);
            if (x0 > x1 || y0 > y1) continue;
        }
        // (P_o . P_t) summed in world-axis order, (x + y) + z, with the face's selection hoisted out of the loops (left
        // inside, the switch on the face costs ~20 scalar branches per pair): exactly one axis carries fx, the others
        // are constant or carry fy, so  P_o . P_t = (c fx + b) + e  with b = bk + br fy, e = ek + er fy  (adding an
        // exact 0 changes nothing, and the commuted first sum of faces 0 / 1 is the same number)
        float dc, bk, br, ek, er;
        switch (s) {
            case 0: dc = -Po.z; bk = Po.x; br = -Po.y; ek = 0.f; er = 0.f; break;     // ( 1, -fy, -fx)
            case 1: dc = Po.z; bk = -Po.x; br = -Po.y; ek = 0.f; er = 0.f; break;     // (-1, -fy,  fx)
            case 2: dc = Po.x; bk = Po.y; br = 0.f; ek = 0.f; er = Po.z; break;       // (fx,  1,  fy)
            case 3: dc = Po.x; bk = -Po.y; br = 0.f; ek = 0.f; er = -Po.z; break;     // (fx, -1, -fy)
            case 4: dc = Po.x; bk = 0.f; br = -Po.y; ek = Po.z; er = 0.f; break;      // (fx, -fy,  1)
            default: dc = -Po.x; bk = 0.f; br = -Po.y; ek = -Po.z; er = 0.f; break;   // (-fx, -fy, -1)
        }
        auto cosine = [&](float fx, float fy, float b, float e) {
            return ((dc * fx + b) + e) * (ro * __builtin_amdgcn_rsqf((fx * fx + fy * fy) + 1.f));
        };
        if (TPO == 64) {
            // one output per wave: the lanes tile the box in 16 x 4 blocks (a box is ~30 texels wide at the 256^2 level:
            // 64 lanes along one row would leave half of them idle), two blocks per step
            const int lx = sub & 15, ly = sub >> 4;
            for (int yb = y0; yb <= y1; yb += 8) {
                const int ya = yb + ly, yc = ya + 4;
                const bool ra = ya <= y1, rc = yc <= y1;
                const float fya = org + step * (float)ya, fyc = org + step * (float)yc;
                const float aya = (BWD || !ra) ? 0.f : 0.25f * ax[ya], ayc = (BWD || !rc) ? 0.f : 0.25f * ax[yc];
                const float ba = bk + br * fya, ea = ek + er * fya, bc = bk + br * fyc, ec = ek + er * fyc;
                const int r0 = N * N * s + N * (ra ? ya : y0), r1 = N * N * s + N * (rc ? yc : y0);
                for (int xb = x0; xb <= x1; xb += 16) {
                    const int x = xb + lx;
                    const bool cx = x <= x1;
                    const int xs = cx ? x : x0;
                    const float fx = org + step * (float)xs;
                    const float da = cosine(fx, fya, ba, ea), db = cosine(fx, fyc, bc, ec);
                    const float* pa = in + (uint32_t)(r0 + xs) * (uint32_t)IC;  // 32-bit offset from the uniform base
                    const float* pb = in + (uint32_t)(r1 + xs) * (uint32_t)IC;
                    const float ca0 = pa[0], ca1 = pa[1], ca2 = pa[2], cb0 = pb[0], cb1 = pb[1], cb2 = pb[2];
                    const float axx = BWD ? 0.f : ax[xs];
                    if (cx && ra) accumulate(da, axx * aya, ca0, ca1, ca2);
                    if (cx && rc) accumulate(db, axx * ayc, cb0, cb1, cb2);
                }
            }
            continue;
        }
        // the group's lanes stride along x; two rows per step so that two colour loads are in flight
        for (int y = y0; y <= y1; y += 2) {
            const bool two = y + 1 <= y1;
            const int r0 = N * N * s + N * y, r1 = two ? r0 + N : r0;
            const float fya = org + step * (float)y, fyb = fya + step;
            const float aya = BWD ? 0.f : 0.25f * ax[y], ayb = BWD ? 0.f : 0.25f * ax[two ? y + 1 : y];
            const float ba = bk + br * fya, ea = ek + er * fya, bb = bk + br * fyb, eb = ek + er * fyb;
            for (int x = x0 + sub; x <= x1; x += TPO) {
                const float fx = org + step * (float)x;
                const float da = cosine(fx, fya, ba, ea), db = cosine(fx, fyb, bb, eb);
                const float* pa = in + (uint32_t)(r0 + x) * (uint32_t)IC;  // 32-bit offset from the uniform base
                const float* pb = in + (uint32_t)(r1 + x) * (uint32_t)IC;
                const float ca0 = pa[0], ca1 = pa[1], ca2 = pa[2], cb0 = pb[0], cb1 = pb[1], cb2 = pb[2];
                const float axx = BWD ? 0.f : ax[x];
                accumulate(da, axx * aya, ca0, ca1, ca2);
                if (two) accumulate(db, axx * ayb, cb0, cb1, cb2);
            }
        }
    }
    if (TPO > 1) {
#pragma unroll
        for (int d = TPO / 2; d >= 1; d >>= 1) {
            a0 += __shfl_xor(a0, d); a1 += __shfl_xor(a1, d); a2 += __shfl_xor(a2, d); ws += __shfl_xor(ws, d);
        }
    }
    if (!live || sub != 0) return;
    if (BWD) {
        const float k = ax[ox] * ax[oy] * 0.25f;
        out[3 * (size_t)o] = a0 * k; out[3 * (size_t)o + 1] = a1 * k; out[3 * (size_t)o + 2] = a2 * k;
    } else {
        out[4 * (size_t)o] = a0; out[4 * (size_t)o + 1] = a1; out[4 * (size_t)o + 2] = a2; out[4 * (size_t)o + 3] = ws;
    }
}

// Mid-size lobes (tens to a couple of thousand pairs per output: the 256^2 and 128^2 levels of a 512^2 light, 80 % of the
// prefilter's time): ONE WAVE PER 8x8 TILE OF OUTPUTS, lane = output.  The cones of neighbouring outputs overlap almost
// completely, so the wave walks the UNION of its 64 boxes on every face it reaches -- a wave-uniform loop -- and every
// texel's data (colour, fx, 1/|P_t|, area) is loaded and derived ONCE per wave by the lane that owns its column and
// broadcast through LDS (v_readlane was tried first: 6 per texel with their SGPR hazards ran 3x slower than the arithmetic):
// per (output, texel) pair 5 operations for L . V from the lane's own constants plus the cone test, and inside the
// cone the 13 of the weight and the sums.  The 16-lanes-per-output form above spends three times
// that: addressing and conversions per pair, a 16-lane group that covers a 27-texel row in two steps, row and face
// set-up repeated by every group, and a shuffle reduction per output.  L . V is formed by the same expression from the
// same numbers (bit-identical, so forward and backward accept the same pairs); only the order of each output's sum differs
// (row-major over the union box, one lane).
// SPLIT waves share a tile when the level has too few tiles to fill the chip (128^2: 1536): wave k takes every SPLIT-th row
// of the union box and the partial sums are added in wave order at the end.
template <bool BWD, int SPLIT>
__global__ void __launch_bounds__(64 * SPLIT) specular_tile_kernel(int N, float roughness, float cos_cut, const float* __restrict__ ax,
                                                                   const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles = N >> 3;
    const int os = blockIdx.x / (tiles * tiles), tt = blockIdx.x - os * tiles * tiles;
    const int ox = (tt % tiles) * 8 + (lane & 7), oy = (tt / tiles) * 8 + (lane >> 3);
    const int o = (os * N + oy) * N + ox;
    const V3 Vo = texel_dir(ox, oy, os, N);
    const float alpha = roughness * roughness, alpha_sqr = alpha * alpha;
    const float sin_t = sqrtf(fmaxf(1.f - cos_cut * cos_cut, 0.f));
    constexpr int IC = BWD ? 4 : 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ws = 0.f;
    const float pi_inv_a2 = alpha_sqr / 3.14159265358979323846f, a2m1h = 0.5f * (alpha_sqr - 1.f);
    const float step = 2.0f / (float)N, org = 1.0f / (float)N - 1.0f;
    const float fxo = org + step * (float)ox, fyo = org + step * (float)oy;
    const V3 Po = face_point(os, fxo, fyo);
    const float ro = __builtin_amdgcn_rsqf((fxo * fxo + fyo * fyo) + 1.f);
    __shared__ float4 s_geo_[SPLIT][64];  // per texel of the segment: fx, 1 / |P_t|, area / 4, colour.x
    __shared__ float2 s_col_[SPLIT][64];  // colour.y, colour.z
    float4* const s_geo = s_geo_[wv];
    float2* const s_col = s_col_[wv];
    // LDS operations of one wave execute in order: what orders a wave's parking writes and broadcast reads is only the
    // compiler (the waves of a split tile run different trip counts: no workgroup barrier inside the loops)
    auto wave_fence = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    const bool pos_cut = cos_cut > 0.f;
    // The union of the tile's 64 cone boxes on every face, found ONCE per workgroup: the waves of the tile share the faces
    // (a box costs two cone_range calls: atan2 / asin / tan, ~700 instructions -- as much as 35 texels of the walk).
    __shared__ int s_box[6][4];
    for (int s = wv; s < 6; s += SPLIT) {
        // this lane's box on face s, exactly as specular_kernel finds it (empty: x0 > x1)
        const V3 c = to_face_frame(s, Vo);
        int x0 = N, x1 = -1, y0 = N, y1 = -1;
        if (cos_cut <= 0.70710678f) { x0 = 0; x1 = N - 1; y0 = 0; y1 = N - 1; }
        else if (c.z > -sin_t && !(fmaxf(fabsf(c.x), fabsf(c.y)) - c.z > 1.41421356f * sin_t + 2e-3f)) {
            cone_range(c.x, c.z, sin_t, N, x0, x1);
            cone_range(c.y, c.z, sin_t, N, y0, y1);
            if (x0 > x1 || y0 > y1) { x0 = N; x1 = -1; y0 = N; y1 = -1; }
        }
        if (__builtin_amdgcn_ballot_w64(x0 <= x1) != 0ull) {  // some output of the tile reaches this face
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                x0 = min(x0, __shfl_xor(x0, d)); y0 = min(y0, __shfl_xor(y0, d));
                x1 = max(x1, __shfl_xor(x1, d)); y1 = max(y1, __shfl_xor(y1, d));
            }
        }
        if (lane == 0) { s_box[s][0] = x0; s_box[s][1] = x1; s_box[s][2] = y0; s_box[s][3] = y1; }
    }
    gs2m_sync();
    for (int s = 0; s < 6; s++) {
        const int X0 = __builtin_amdgcn_readfirstlane(s_box[s][0]), X1 = __builtin_amdgcn_readfirstlane(s_box[s][1]);
        const int Y0 = __builtin_amdgcn_readfirstlane(s_box[s][2]), Y1 = __builtin_amdgcn_readfirstlane(s_box[s][3]);
        if (X0 > X1 || Y0 > Y1) continue;
        float dc, bk, br, ek, er;  // P_o . P_t = (dc fx + (bk + br fy)) + (ek + er fy), see specular_kernel
        switch (s) {
            case 0: dc = -Po.z; bk = Po.x; br = -Po.y; ek = 0.f; er = 0.f; break;
            case 1: dc = Po.z; bk = -Po.x; br = -Po.y; ek = 0.f; er = 0.f; break;
            case 2: dc = Po.x; bk = Po.y; br = 0.f; ek = 0.f; er = Po.z; break;
            case 3: dc = Po.x; bk = -Po.y; br = 0.f; ek = 0.f; er = -Po.z; break;
            case 4: dc = Po.x; bk = 0.f; br = -Po.y; ek = Po.z; er = 0.f; break;
            default: dc = -Po.x; bk = 0.f; br = -Po.y; ek = -Po.z; er = 0.f; break;
        }
        // The union box in segments of up to 64 texels of one row.  The lane that owns a texel's column loads its colour
        // (one segment ahead of its use), derives fx, 1 / |P_t| and the area factor, and parks the eight floats in LDS; the
        // evaluation reads them back with wave-uniform addresses (broadcasts).  One wave per workgroup, LDS in order.
        int y = Y0 + wv, xb = X0;
        if (y > Y1) continue;
        auto fetch = [&](int yy, int xx, float& q0, float& q1, float& q2) {
            const int xt = min(xx + lane, X1);  // lanes past the end re-read the last texel; never evaluated
            const float* p = in + (uint32_t)(N * N * s + N * yy + xt) * (uint32_t)IC;
            q0 = p[0]; q1 = p[1]; q2 = p[2];
        };
        float q0, q1, q2;
        fetch(y, xb, q0, q1, q2);
        while (true) {
            const float fy = org + step * (float)y;
            {
                const int xt = min(xb + lane, X1);
                const float fxt = org + step * (float)xt;
                const float rt = __builtin_amdgcn_rsqf((fxt * fxt + fy * fy) + 1.f);
                const float at = BWD ? 0.f : ax[xt] * (0.25f * ax[y]);
                s_geo[lane] = make_float4(fxt, rt, at, q0);
                s_col[lane] = make_float2(q1, q2);
            }
            wave_fence();
            int yn = y, xn = xb + 64;
            if (xn > X1) { xn = X0; yn = y + SPLIT; }
            const bool more = yn <= Y1;
            if (more) fetch(yn, xn, q0, q1, q2);
            const float b = bk + br * fy, e = ek + er * fy;
            const int w = __builtin_amdgcn_readfirstlane(min(64, X1 - xb + 1));
            auto pair = [&](const float4 g, const float2 cl) {
                const float d = ((dc * g.x + b) + e) * (ro * g.y);
                if (d >= cos_cut) {
                    const float den = __builtin_fmaf(1.f + d, a2m1h, 1.f);
                    // inside the cone d >= cos_cut; for a positive cutoff (every lobe narrower than a hemisphere) max(d, 0) = d
                    const float k = (pos_cut ? d : fmaxf(d, 0.f)) * pi_inv_a2 * __builtin_amdgcn_rcpf(den * den);
                    const float wgt = BWD ? k : k * g.z;
                    a0 = __builtin_fmaf(wgt, g.w, a0); a1 = __builtin_fmaf(wgt, cl.x, a1); a2 = __builtin_fmaf(wgt, cl.y, a2);
                    ws += wgt;
                }
            };
            // four texels per step, all eight broadcast reads in front of the arithmetic: a read issued inside the cone
            // test's branch waits out the LDS latency there
            int l = 0;
            for (; l + 3 < w; l += 4) {
                const float4 g0 = s_geo[l], g1 = s_geo[l + 1], g2 = s_geo[l + 2], g3 = s_geo[l + 3];
                const float2 c0 = s_col[l], c1 = s_col[l + 1], c2 = s_col[l + 2], c3 = s_col[l + 3];
                pair(g0, c0);
                pair(g1, c1);
                pair(g2, c2);
                pair(g3, c3);
            }
            for (; l < w; l++) pair(s_geo[l], s_col[l]);
            wave_fence();
            if (!more) break;
            y = yn; xb = xn;
        }
    }
    if (SPLIT > 1) {  // partial sums of waves 1 .. SPLIT - 1, added in wave order
        __shared__ float4 s_part[SPLIT][64];
        s_part[wv][lane] = make_float4(a0, a1, a2, ws);
        gs2m_sync();
        if (wv != 0) return;
#pragma unroll
        for (int k = 1; k < SPLIT; k++) {
            const float4 q = s_part[k][lane];
            a0 += q.x; a1 += q.y; a2 += q.z; ws += q.w;
        }
    }
    if (BWD) {
        const float k = ax[ox] * ax[oy] * 0.25f;
        out[3 * (size_t)o] = a0 * k; out[3 * (size_t)o + 1] = a1 * k; out[3 * (size_t)o + 2] = a2 * k;
    } else {
        reinterpret_cast<float4*>(out)[o] = make_float4(a0, a1, a2, ws);
    }
}

// the division by the weight sum (render_utils/ops.py:403) and its backward as part of the operator: in PyTorch the slice /
// divide and their autograd nodes are ~13 launches per level, six levels per view
__global__ void __launch_bounds__(256) specular_normalize_kernel(int n, const float4* __restrict__ raw, float* __restrict__ out3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r = raw[i];
    out3[3 * (size_t)i] = r.x / r.w; out3[3 * (size_t)i + 1] = r.y / r.w; out3[3 * (size_t)i + 2] = r.z / r.w;
}
__global__ void __launch_bounds__(256) specular_prescale_kernel(int n, const float4* __restrict__ raw, const float* __restrict__ g3,
                                                                float4* __restrict__ g4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float w = raw[i].w;  // d(col / w)/d col = 1 / w; w does not depend on the cubemap
    g4[i] = make_float4(g3[3 * (size_t)i] / w, g3[3 * (size_t)i + 1] / w, g3[3 * (size_t)i + 2] / w, 0.f);
}

template <bool BWD>
int launch_specular(int res, float roughness, float cos_cut, const float* tab, const float* in, float* out, hipStream_t s) {
    const long long total = 6LL * res * res;
    // expected texel pairs per output: the cone's share of the sphere
    const double pairs = 0.5 * (1.0 - (double)cos_cut) * (double)total;
    // one output per wave for wide lobes and for small levels (few outputs: 16 lanes each would leave the chip empty);
    // measured: 64 lanes per output LOSE on the 256^2 / 128^2 levels (1.37 vs 0.88 ms, 0.55 vs 0.41 ms)
    if (pairs >= 48.0 && pairs < 4096.0 && res >= 64 && res % 8 == 0 && ((uintptr_t)out & 15) == 0) {
        const int tiles = 6 * (res / 8) * (res / 8);
        if (tiles >= 1024) specular_tile_kernel<BWD, 4><<<tiles, 256, 0, s>>>(res, roughness, cos_cut, tab, in, out);
        else specular_tile_kernel<BWD, 8><<<tiles, 512, 0, s>>>(res, roughness, cos_cut, tab, in, out);
    } else if (pairs >= 2048.0 || (pairs >= 48.0 && total <= 8192)) specular_kernel<BWD, 64><<<(unsigned)((total * 64 + 255) / 256), 256, 0, s>>>(res, roughness, cos_cut, tab, in, out);
    else if (pairs >= 48.0) specular_kernel<BWD, 16><<<(unsigned)((total * 16 + 255) / 256), 256, 0, s>>>(res, roughness, cos_cut, tab, in, out);
    else specular_kernel<BWD, 1><<<(unsigned)((total + 255) / 256), 256, 0, s>>>(res, roughness, cos_cut, tab, in, out);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

}  // namespace

extern "C" {

int gs2m_diffuse_cubemap_forward(int res, const float* cubemap, float* out, void* stream) {
    if (res < 1 || res > 1024 || !cubemap || !out) return GS2M_ERR_INVALID_ARG;
    diffuse_kernel<false><<<6 * res * res, 64, 0, (hipStream_t)stream>>>(res, cubemap, out);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_diffuse_cubemap_backward(int res, const float* dL_dout, float* dL_dcubemap, void* stream) {
    if (res < 1 || res > 1024 || !dL_dout || !dL_dcubemap) return GS2M_ERR_INVALID_ARG;
    diffuse_kernel<true><<<6 * res * res, 64, 0, (hipStream_t)stream>>>(res, dL_dout, dL_dcubemap);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_cubemap_texel_table(int res, float* table, void* stream) {
    if (res < 1 || res > 4096 || !table) return GS2M_ERR_INVALID_ARG;
    axis_area_kernel<<<(res + 255) / 256, 256, 0, (hipStream_t)stream>>>(res, table);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_specular_cubemap_forward(int res, float roughness, float costheta_cutoff, const float* texel_table, const float* cubemap,
                                  float* out, void* stream) {
    if (res < 1 || res > 4096 || !texel_table || !cubemap || !out) return GS2M_ERR_INVALID_ARG;
    return launch_specular<false>(res, roughness, costheta_cutoff, texel_table, cubemap, out,
                                  (hipStream_t)stream);
}

int gs2m_specular_cubemap_backward(int res, float roughness, float costheta_cutoff, const float* texel_table, const float* dL_dout,
                                   float* dL_dcubemap, void* stream) {
    if (res < 1 || res > 4096 || !texel_table || !dL_dout || !dL_dcubemap) return GS2M_ERR_INVALID_ARG;
    return launch_specular<true>(res, roughness, costheta_cutoff, texel_table, dL_dout, dL_dcubemap,
                                 (hipStream_t)stream);
}

int gs2m_specular_cubemap_normalized_forward(int res, float roughness, float costheta_cutoff, const float* texel_table,
                                             const float* cubemap, float* raw, float* out, void* stream) {
    if (res < 1 || res > 4096 || !texel_table || !cubemap || !raw || !out || ((uintptr_t)raw & 15)) return GS2M_ERR_INVALID_ARG;
    const int rc = launch_specular<false>(res, roughness, costheta_cutoff, texel_table, cubemap, raw, (hipStream_t)stream);
    if (rc != GS2M_OK) return rc;
    const int n = 6 * res * res;
    specular_normalize_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, reinterpret_cast<const float4*>(raw), out);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_specular_cubemap_normalized_backward(int res, float roughness, float costheta_cutoff, const float* texel_table,
                                              const float* raw, const float* dL_dout, float* scratch, float* dL_dcubemap, void* stream) {
    if (res < 1 || res > 4096 || !texel_table || !raw || !dL_dout || !scratch || !dL_dcubemap || (((uintptr_t)raw | (uintptr_t)scratch) & 15))
        return GS2M_ERR_INVALID_ARG;
    const int n = 6 * res * res;
    specular_prescale_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, reinterpret_cast<const float4*>(raw), dL_dout,
                                                                             reinterpret_cast<float4*>(scratch));
    if (hipGetLastError() != hipSuccess) return GS2M_ERR_HIP;
    return launch_specular<true>(res, roughness, costheta_cutoff, texel_table, scratch, dL_dcubemap, (hipStream_t)stream);
}

}  // extern "C"
