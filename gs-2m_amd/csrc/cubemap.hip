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
// ~9 pairs, a 32^2 level with a 40-degree lobe 6k outputs of ~700); texel directions and areas come from a per-level
// table the caller caches, so a pair is a 16-byte load, a dot product and, inside the cone, ~10 more operations.
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

__device__ __forceinline__ float ndf_ggx(float alpha_sqr, float cos_theta) {  // cubemap.cu:193-198
    const float c = fminf(fmaxf(cos_theta, 0.f), 1.f);
    const float d = (c * alpha_sqr - c) * c + 1.0f;
    return alpha_sqr / (d * d * 3.14159265358979323846f);
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
// per-texel table: (direction, area), so a pair costs a 16-byte load, a dot product and -- inside the cone -- the lobe:
// for unit vectors V . H = sqrt((1 + L . V) / 2), no half vector needed
__global__ void __launch_bounds__(256) texel_table_kernel(int N, float4* __restrict__ tab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 6 * N * N) return;
    const int s = t / (N * N), y = (t / N) % N, x = t % N;
    const V3 L = texel_dir(x, y, s, N);
    tab[t] = make_float4(L.x, L.y, L.z, texel_area(x, y, N));
}

// forward: o = output texel, gathers inputs (3 channels in, 4 out); backward: o = input texel, gathers output gradients
// (4 channels in, of which the 4th -- d/d wsum -- does not reach the cubemap; 3 out).  TPO lanes share one output texel
// and stride over the box of every face (wide lobes on small levels would otherwise leave the chip empty).
template <bool BWD, int TPO>
__global__ void __launch_bounds__(256) specular_kernel(int N, float roughness, float cos_cut, const float4* __restrict__ tab,
                                                       const float* __restrict__ in, float* __restrict__ out) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 6 * N * N;
    const int o_raw = gt / TPO, sub = gt % TPO;
    const bool live = o_raw < total;
    const int o = live ? o_raw : total - 1;  // keep whole groups converged for the reduction
    const float4 T = tab[o];
    const V3 Vo = v3(T.x, T.y, T.z);
    const float alpha = roughness * roughness, alpha_sqr = alpha * alpha;
    const float sin_t = sqrtf(fmaxf(1.f - cos_cut * cos_cut, 0.f));
    constexpr int IC = BWD ? 4 : 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ws = 0.f;
    auto pair = [&](int t) {
        const float4 L = tab[t];
        const float d = L.x * Vo.x + L.y * Vo.y + L.z * Vo.z;
        if (d >= cos_cut) {
            const float k = fmaxf(d, 0.f) * ndf_ggx(alpha_sqr, sqrtf(fmaxf(0.5f + 0.5f * d, 0.f)));
            const float w = BWD ? k : k * L.w * 0.25f;
            const float* p = in + (size_t)t * IC;
            a0 += w * p[0]; a1 += w * p[1]; a2 += w * p[2];
            ws += w;
        }
    };
    for (int s = 0; s < 6; s++) {
        const V3 c = to_face_frame(s, Vo);
        int x0, x1, y0, y1;
        if (cos_cut <= 0.70710678f) { x0 = 0; x1 = N - 1; y0 = 0; y1 = N - 1; }  // half-angle >= 45 deg: no useful box
        else {
            if (c.z <= -sin_t) continue;  // the whole cone is behind the face's plane
            cone_range(c.x, c.z, sin_t, N, x0, x1);
            cone_range(c.y, c.z, sin_t, N, y0, y1);
            if (x0 > x1 || y0 > y1) continue;
        }
        const int base = N * N * s;
        if (TPO == 1) {
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) pair(base + x + N * y);
        } else {
            const int wdt = x1 - x0 + 1, cnt = wdt * (y1 - y0 + 1);
            const float inv = 1.0f / (float)wdt;
            for (int idx = sub; idx < cnt; idx += TPO) {
                const int row = (int)(((float)idx + 0.5f) * inv);  // exact for idx < 2^22
                pair(base + x0 + (idx - row * wdt) + N * (y0 + row));
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
        const float k = T.w * 0.25f;
        out[3 * (size_t)o] = a0 * k; out[3 * (size_t)o + 1] = a1 * k; out[3 * (size_t)o + 2] = a2 * k;
    } else {
        out[4 * (size_t)o] = a0; out[4 * (size_t)o + 1] = a1; out[4 * (size_t)o + 2] = a2; out[4 * (size_t)o + 3] = ws;
    }
}

template <bool BWD>
int launch_specular(int res, float roughness, float cos_cut, const float4* tab, const float* in, float* out, hipStream_t s) {
    const long long total = 6LL * res * res;
    // expected texel pairs per output: the cone's share of the sphere
    const double pairs = 0.5 * (1.0 - (double)cos_cut) * (double)total;
    if (pairs >= 2048.0) specular_kernel<BWD, 64><<<(unsigned)((total * 64 + 255) / 256), 256, 0, s>>>(res, roughness, cos_cut, tab, in, out);
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
    if (res < 1 || res > 4096 || !table || ((uintptr_t)table & 15)) return GS2M_ERR_INVALID_ARG;
    texel_table_kernel<<<(6 * res * res + 255) / 256, 256, 0, (hipStream_t)stream>>>(res, reinterpret_cast<float4*>(table));
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_specular_cubemap_forward(int res, float roughness, float costheta_cutoff, const float* texel_table, const float* cubemap,
                                  float* out, void* stream) {
    if (res < 1 || res > 4096 || !texel_table || ((uintptr_t)texel_table & 15) || !cubemap || !out) return GS2M_ERR_INVALID_ARG;
    return launch_specular<false>(res, roughness, costheta_cutoff, reinterpret_cast<const float4*>(texel_table), cubemap, out,
                                  (hipStream_t)stream);
}

int gs2m_specular_cubemap_backward(int res, float roughness, float costheta_cutoff, const float* texel_table, const float* dL_dout,
                                   float* dL_dcubemap, void* stream) {
    if (res < 1 || res > 4096 || !texel_table || ((uintptr_t)texel_table & 15) || !dL_dout || !dL_dcubemap) return GS2M_ERR_INVALID_ARG;
    return launch_specular<true>(res, roughness, costheta_cutoff, reinterpret_cast<const float4*>(texel_table), dL_dout, dL_dcubemap,
                                 (hipStream_t)stream);
}

}  // extern "C"
