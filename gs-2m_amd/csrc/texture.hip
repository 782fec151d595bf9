// Texture lookups of the deferred PBR stage for gfx950 (the ROCm replacement of the
// `nvdiffrast.torch.texture` calls in pbr/shade.py:150-190 and pbr/light.py:43-48, 111-115).
//
// Three modes, the ones the reference uses, with nvdiffrast's filtering semantics
// (submodules/nvdiffrast/nvdiffrast/common/textureCUDA.cu):
//   cube map, 'linear', boundary 'cube'                 diffuse irradiance lookup by the normal; cubemap_mip backward
//   cube map, 'linear-mipmap-linear' with an explicit mip stack and a per-pixel mip_level_bias, no uv derivatives
//                                                       (so the level IS the clamped bias, :575-589): specular lookup
//   2-D, 'linear', boundary 'clamp'                     BRDF LUT lookup
// and the backward with respect to the texture (persistent workgroups; runs of equal texels merged inside the wave; the
// small levels accumulate in a workgroup-private LDS copy; every level its own gradient tensor: explicit mips are independent
// inputs there too).  Gradients with respect to uv / the bias are not produced: the reference detaches both
// (pbr/__init__.py:25-43), and the Python mirror refuses inputs that require them.
//
// Semantics restated: a direction picks the face of its largest |component| (ties: z over y over x only when
// strictly larger, :99-121), face coordinates follow the OpenGL convention (the same table as pbr/light.py:13-26),
// texel centres sit at (i + 0.5) / w, bilinear weights come from u * w - 0.5 (:383-384, :417-421).  A texel index
// that leaves the face across ONE edge continues on the neighbouring face (wrapCubeMap, :47-92, table driven there;
// here the texel centre is folded around the cube edge in exact integer arithmetic, which yields the same
// neighbour); one that leaves across a CORNER does not exist and takes the average of the other three
// (fetchQuad :590-607; its weight goes to them in thirds in the backward, accumQuad :616-631).  A non-finite
// direction gives zero output and no gradient (:116-117).
//
// One thread per output pixel; channels-last textures (6, w, w, C) / (H, W, C), C <= 4.  The textures are small
// (<= 19 MB) and stay in L2 / MALL; the forward is bound by the streaming uv read and output write, the backward by
// the atomic rate.
#include "common.h"
#include "../../include/gs2m_texture.h"
#include "../../include/gs2m_pbr.h"

namespace {

struct CubeTexel {  // texel address of one corner of the bilinear footprint; f < 0: does not exist
    int f, x, y;
};

// Texel (ix, iy) of face f at width w, with at most one texel of overshoot per axis, -> the texel it denotes.
__device__ __forceinline__ CubeTexel cube_fold(int f, int ix, int iy, int w) {
    const bool ox = ix < 0 || ix >= w, oy = iy < 0 || iy >= w;
    if (!ox && !oy) return {f, ix, iy};
    if (ox && oy) return {-1, 0, 0};
    // doubled, centred face coordinates: in range they are odd integers in (-w, w); the cube has half-width w
    int X = 2 * ix + 1 - w, Y = 2 * iy + 1 - w;
    int p[3];
    switch (f) {  // point on the face plane (pbr/light.py:13-26)
        case 0: p[0] = w; p[1] = -Y; p[2] = -X; break;
        case 1: p[0] = -w; p[1] = -Y; p[2] = X; break;
        case 2: p[0] = X; p[1] = w; p[2] = Y; break;
        case 3: p[0] = X; p[1] = -w; p[2] = -Y; break;
        case 4: p[0] = X; p[1] = -Y; p[2] = w; break;
        default: p[0] = -X; p[1] = -Y; p[2] = -w; break;
    }
    const int major = f >> 1;
    // fold the overshooting component around the edge: it becomes the new major axis at +-w, and the old major
    // axis retreats from the edge by the overshoot (1 in these units, i.e. half a texel: the neighbour's first row)
    int g = -1;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (a != major && (p[a] > w || p[a] < -w)) {
            const int over = (p[a] > 0 ? p[a] : -p[a]) - w;
            p[major] += p[major] > 0 ? -over : over;
            p[a] = p[a] > 0 ? w : -w;
            g = 2 * a + (p[a] < 0 ? 1 : 0);
        }
    }
    switch (g) {  // inverse of the table above on the neighbouring face
        case 0: X = -p[2]; Y = -p[1]; break;
        case 1: X = p[2]; Y = -p[1]; break;
        case 2: X = p[0]; Y = p[2]; break;
        case 3: X = p[0]; Y = -p[2]; break;
        case 4: X = p[0]; Y = -p[1]; break;
        default: X = -p[0]; Y = -p[1]; break;
    }
    return {g, (X + w - 1) >> 1, (Y + w - 1) >> 1};
}

// direction -> face and texture coordinates in [0, 1]; -1 for a non-finite result
__device__ __forceinline__ int cube_index(float x, float y, float z, float& u, float& v) {
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    int f;
    float c, s, t;  // major component and the two that become (u, v)
    if (az > fmaxf(ax, ay)) { f = 4; c = z; s = x; t = y; }
    else if (ay > ax)       { f = 2; c = y; s = x; t = z; }
    else                    { f = 0; c = x; s = z; t = y; }
    if (c < 0.f) f += 1;
    const float m = 0.5f / fabsf(c);
    const float m0 = (f == 0 || f == 5) ? -m : m;
    const float m1 = (f != 2) ? -m : m;
    u = s * m0 + 0.5f;
    v = t * m1 + 0.5f;
    if (!isfinite(u) || !isfinite(v)) return -1;
    u = fminf(fmaxf(u, 0.f), 1.f);
    v = fminf(fmaxf(v, 0.f), 1.f);
    return f;
}

struct Footprint {  // four texel offsets (in texels from the level base; < 0: missing) and the bilinear fractions
    int t[4];
    float fu, fv;
    bool corner;
};

__device__ __forceinline__ Footprint cube_footprint(float x, float y, float z, int w) {
    Footprint F;
    float u, v;
    const int f = cube_index(x, y, z, u, v);
    if (f < 0) {
        F.t[0] = F.t[1] = F.t[2] = F.t[3] = -1; F.fu = F.fv = 0.f; F.corner = false;
        return F;
    }
    u = u * (float)w - 0.5f;
    v = v * (float)w - 0.5f;
    const int iu0 = (int)floorf(u), iv0 = (int)floorf(v);
    F.fu = u - (float)iu0;
    F.fv = v - (float)iv0;
    bool missing = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const CubeTexel c = cube_fold(f, iu0 + (k & 1), iv0 + (k >> 1), w);
        F.t[k] = c.f < 0 ? -1 : c.x + w * (c.y + w * c.f);
        missing |= c.f < 0;
    }
    F.corner = missing;
    return F;
}

__device__ __forceinline__ Footprint clamp2d_footprint(float u, float v, int w, int h) {  // textureCUDA.cu:395-421
    Footprint F;
    u = u * (float)w - 0.5f;
    v = v * (float)h - 0.5f;
    u = fminf(fmaxf(u, 0.f), (float)w - 1.f);
    v = fminf(fmaxf(v, 0.f), (float)h - 1.f);
    const bool cu = (u == 0.f || u == (float)w - 1.f), cv = (v == 0.f || v == (float)h - 1.f);
    const int iu0 = (int)floorf(u), iv0 = (int)floorf(v);
    const int iu1 = iu0 + (cu ? 0 : 1), iv1 = iv0 + (cv ? 0 : 1);
    F.fu = u - (float)iu0;
    F.fv = v - (float)iv0;
    F.t[0] = iu0 + w * iv0; F.t[1] = iu1 + w * iv0; F.t[2] = iu0 + w * iv1; F.t[3] = iu1 + w * iv1;
    F.corner = false;
    return F;
}

__device__ __forceinline__ float lerpf(float a, float b, float c) { return a + c * (b - a); }

template <int C>
__device__ __forceinline__ void sample(const float* __restrict__ tex, const Footprint& F, float (&out)[C]) {
    float a[4][C];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < C; c++) a[k][c] = F.t[k] >= 0 ? tex[(size_t)F.t[k] * C + c] : 0.f;
    if (F.corner) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float avg = (a[0][c] + a[1][c] + a[2][c] + a[3][c]) * 0.33333333f;  // the missing one contributed 0
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (F.t[k] < 0) a[k][c] = avg;
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) out[c] = lerpf(lerpf(a[0][c], a[1][c], F.fu), lerpf(a[2][c], a[3][c], F.fu), F.fv);
}

// Backward accumulation.  The four corner weights of one footprint (corner case folded in):
__device__ __forceinline__ void footprint_weights(const Footprint& F, float (&wt)[4]) {
    wt[0] = (1.f - F.fu) * (1.f - F.fv); wt[1] = F.fu * (1.f - F.fv); wt[2] = (1.f - F.fu) * F.fv; wt[3] = F.fu * F.fv;
    if (F.corner) {
        float cb = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (F.t[k] < 0) cb = wt[k];
        cb *= 0.33333333f;
#pragma unroll
        for (int k = 0; k < 4; k++) wt[k] += cb;
    }
}

// Neighbouring pixels of a smooth image land on the same texel, so the 64 lanes of a wave would issue atomics to a
// handful of addresses, which serialise.  Runs of equal keys among the 16 lanes of a DPP row are summed first (a
// segmented scan, 4 row_shr steps) and only the last lane of each run issues the atomic.  Must be called by all
// lanes of the wave; key < 0 = nothing to add.
template <int C, int D>
__device__ __forceinline__ void run_merge_step(int lr, int& head, float (&v)[C]) {
    const int hu = __builtin_amdgcn_update_dpp(1, head, 0x110 + D, 0xF, 0xF, false);  // row_shr:D
    float vu[C];
#pragma unroll
    for (int c = 0; c < C; c++) vu[c] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[c]), 0x110 + D, 0xF, 0xF, false));
    if (lr >= D && !head) {
#pragma unroll
        for (int c = 0; c < C; c++) v[c] += vu[c];
        head = hu;
    }
}

template <int C>
__device__ __forceinline__ bool run_merge(int key, float (&v)[C]) {
    const int lr = threadIdx.x & 15;
    const int prev = __builtin_amdgcn_update_dpp(-2, key, 0x111, 0xF, 0xF, false);  // row_shr:1
    const int next = __builtin_amdgcn_update_dpp(-2, key, 0x101, 0xF, 0xF, false);  // row_shl:1
    int head = (lr == 0 || prev != key) ? 1 : 0;
    run_merge_step<C, 1>(lr, head, v);
    run_merge_step<C, 2>(lr, head, v);
    run_merge_step<C, 4>(lr, head, v);
    run_merge_step<C, 8>(lr, head, v);
    return key >= 0 && (lr == 15 || next != key);  // this lane holds its run's sum and issues the add
}

// A combining table in LDS in front of the global atomics of the levels that are too large to privatise: key = (level, texel),
// open addressing, 8 probes, then straight to memory; drained after every pixel tile.  A 32 x 32 tile of a smooth image puts its
// ~8000 contributions on a few hundred distinct texels.
constexpr int COMB_HT = 2048;

template <int C>
struct CombineTable {
    int* key;     // [COMB_HT], -1 = empty
    float* val;   // [COMB_HT][C]
    __device__ __forceinline__ void clear(int tid, int nthreads) {
        for (int k = tid; k < COMB_HT; k += nthreads) {
            key[k] = -1;
#pragma unroll
            for (int c = 0; c < C; c++) val[C * k + c] = 0.f;
        }
    }
    __device__ __forceinline__ void add(int level, int texel, float* gptr, const float (&v)[C]) {
        const int k = (level << 24) | texel;
        unsigned h = ((unsigned)k * 2654435761u) >> 21;  // 11 bits
#pragma unroll 1
        for (int probe = 0; probe < 8; probe++) {
            const int old = atomicCAS(&key[h], -1, k);
            if (old == -1 || old == k) {
#pragma unroll
                for (int c = 0; c < C; c++) atomicAdd(&val[C * h + c], v[c]);
                return;
            }
            h = (h + 1) & (COMB_HT - 1);
        }
#pragma unroll
        for (int c = 0; c < C; c++) unsafeAtomicAdd(&gptr[(size_t)texel * C + c], v[c]);
    }
    template <class GradOf>
    __device__ __forceinline__ void drain(int tid, int nthreads, GradOf grad_of) {  // all threads of the workgroup
        __syncthreads();
        for (int k = tid; k < COMB_HT; k += nthreads) {
            const int kk = key[k];
            if (kk >= 0) {
                float* g = grad_of(kk >> 24) + (size_t)(kk & 0xFFFFFF) * C;
#pragma unroll
                for (int c = 0; c < C; c++) { unsafeAtomicAdd(&g[c], val[C * k + c]); val[C * k + c] = 0.f; }
                key[k] = -1;
            }
        }
        __syncthreads();
    }
};

// pixel index of thread t in tile `tile`: 32 x 32 image tiles when the image width is known (the 16 lanes of a DPP row are
// horizontal neighbours), else runs of 1024 consecutive pixels
__device__ __forceinline__ int tile_pixel(int tile, int t, int n, int img_w, bool& valid) {
    if (img_w > 0) {
        const int img_h = n / img_w, tiles_x = (img_w + 31) / 32;
        const int x = (tile % tiles_x) * 32 + (t & 31), y = (tile / tiles_x) * 32 + (t >> 5);
        valid = x < img_w && y < img_h;
        return y * img_w + x;
    }
    const int i = tile * 1024 + t;
    valid = i < n;
    return i;
}
__host__ __device__ __forceinline__ int tile_count(int n, int img_w) {
    return img_w > 0 ? ((img_w + 31) / 32) * ((n / img_w + 31) / 32) : (n + 1023) / 1024;
}

struct MipStack {
    const float* tex[GS2M_TEX_MAX_LEVELS];
    float* grad[GS2M_TEX_MAX_LEVELS];
    int width[GS2M_TEX_MAX_LEVELS];
    int lds_off[GS2M_TEX_MAX_LEVELS];  // backward: float offset of the level's private copy in dynamic LDS, or -1
    int lds_floats;
    int levels;
};

constexpr int TEX_LDS_MAX_WIDTH = 32;      // levels up to 6 x 32 x 32 x C are privatised (98 KB at C = 4, + 24 KB for 16^2)
constexpr int TEX_BWD_THREADS = 1024;

// mode 0: cube linear (level 0 only); 1: cube linear-mipmap-linear by bias; 2: 2-D clamp linear (width x height)
template <int C, int MODE>
__global__ void __launch_bounds__(256) texture_kernel(int n, MipStack M, int height, const float* __restrict__ uv,
                                                      const float* __restrict__ bias, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r[C];
    if (MODE == 2) {
        sample<C>(M.tex[0], clamp2d_footprint(uv[2 * (size_t)i], uv[2 * (size_t)i + 1], M.width[0], height), r);
    } else {
        const float x = uv[3 * (size_t)i], y = uv[3 * (size_t)i + 1], z = uv[3 * (size_t)i + 2];
        int l0 = 0, l1 = 0;
        float fl = 0.f;
        if (MODE == 1) {  // calculateMipLevel with BIAS_ONLY, textureCUDA.cu:575-589
            fl = fminf(fmaxf(bias[i], 0.f), (float)(M.levels - 1));
            l0 = (int)floorf(fl);
            if (fl > 0.f) { l1 = min(l0 + 1, M.levels - 1); fl -= (float)l0; }
        }
        sample<C>(M.tex[l0], cube_footprint(x, y, z, M.width[l0]), r);
        if (MODE == 1 && fl > 0.f) {
            float r1[C];
            sample<C>(M.tex[l1], cube_footprint(x, y, z, M.width[l1]), r1);
#pragma unroll
            for (int c = 0; c < C; c++) r[c] = lerpf(r[c], r1[c], fl);
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) out[(size_t)i * C + c] = r[c];
}

// backward: persistent workgroups (one per CU), grid-stride over the pixels; small levels accumulate in LDS and are
// flushed once per workgroup
template <int C, int MODE>
__global__ void __launch_bounds__(TEX_BWD_THREADS) texture_bwd_kernel(int n, int img_w, MipStack M, int height, const float* __restrict__ uv,
                                                                      const float* __restrict__ bias, const float* __restrict__ dy) {
    extern __shared__ float s_acc[];
    CombineTable<C> T = {reinterpret_cast<int*>(s_acc + M.lds_floats), s_acc + M.lds_floats + COMB_HT};
    for (int k = threadIdx.x; k < M.lds_floats; k += TEX_BWD_THREADS) s_acc[k] = 0.f;
    T.clear(threadIdx.x, TEX_BWD_THREADS);
    __syncthreads();
    // wave-uniform: every lane walks the same number of pixels and footprints; lanes with nothing to add carry key -1
    auto add = [&](bool on, int level, const Footprint& F, const float (&g)[C], float scale) {
        float wt[4];
        footprint_weights(F, wt);
        const int loff = M.lds_off[on ? level : 0];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool live = on && F.t[k] >= 0;
            float v[C];
#pragma unroll
            for (int c = 0; c < C; c++) v[c] = live ? wt[k] * scale * g[c] : 0.f;
            const int key = live ? (level << 24) | F.t[k] : -1;
            if (run_merge<C>(key, v)) {
                if (loff >= 0) {
#pragma unroll
                    for (int c = 0; c < C; c++) atomicAdd(&s_acc[loff + F.t[k] * C + c], v[c]);  // ds_add_f32
                } else {
                    T.add(level, F.t[k], M.grad[level], v);
                }
            }
        }
    };
    const int ntiles = tile_count(n, img_w);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        bool valid;
        const int i = tile_pixel(tile, threadIdx.x, n, img_w, valid);
        const size_t ii = valid ? (size_t)i : 0;
        float g[C];
        bool any = false;
#pragma unroll
        for (int c = 0; c < C; c++) { g[c] = dy[ii * C + c]; any |= g[c] != 0.f; }
        any &= valid;
        if (MODE == 2) {
            add(any, 0, clamp2d_footprint(uv[2 * ii], uv[2 * ii + 1], M.width[0], height), g, 1.f);
        } else {
            const float x = uv[3 * ii], y = uv[3 * ii + 1], z = uv[3 * ii + 2];
            int l0 = 0, l1 = 0;
            float fl = 0.f;
            if (MODE == 1) {
                fl = fminf(fmaxf(bias[ii], 0.f), (float)(M.levels - 1));
                l0 = (int)floorf(fl);
                if (fl > 0.f) { l1 = min(l0 + 1, M.levels - 1); fl -= (float)l0; }
            }
            const bool two = MODE == 1 && fl > 0.f;
            add(any, l0, cube_footprint(x, y, z, M.width[l0]), g, two ? 1.f - fl : 1.f);
            if (MODE == 1) add(any && two, l1, cube_footprint(x, y, z, M.width[l1]), g, fl);
        }
        T.drain(threadIdx.x, TEX_BWD_THREADS, [&](int level) { return M.grad[level]; });
    }
    __syncthreads();
    for (int l = 0; l < M.levels; l++) {
        if (M.lds_off[l] < 0) continue;
        const int cnt = (MODE == 2 ? M.width[l] * height : 6 * M.width[l] * M.width[l]) * C;
        // every workgroup starts somewhere else: they all get here at about the same time, and atomics to one address serialise
        const int start = (int)(((long long)blockIdx.x * cnt) / gridDim.x);
        for (int k0 = threadIdx.x; k0 < cnt; k0 += TEX_BWD_THREADS) {
            int k = k0 + start;
            if (k >= cnt) k -= cnt;
            const float v = s_acc[M.lds_off[l] + k];
            if (v != 0.f) unsafeAtomicAdd(&M.grad[l][k], v);
        }
    }
}

template <int MODE>
int launch_fwd(int C, int n, const MipStack& M, int height, const float* uv, const float* bias, float* out, hipStream_t s) {
    const dim3 grid((n + 255) / 256), block(256);
    switch (C) {
        case 1: texture_kernel<1, MODE><<<grid, block, 0, s>>>(n, M, height, uv, bias, out); break;
        case 2: texture_kernel<2, MODE><<<grid, block, 0, s>>>(n, M, height, uv, bias, out); break;
        case 3: texture_kernel<3, MODE><<<grid, block, 0, s>>>(n, M, height, uv, bias, out); break;
        case 4: texture_kernel<4, MODE><<<grid, block, 0, s>>>(n, M, height, uv, bias, out); break;
        default: return GS2M_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

template <int C, int MODE>
int launch_bwd_c(int n, int img_w, MipStack& M, int height, const float* uv, const float* bias, const float* dy, hipStream_t s) {
    // private LDS copies for the small levels (2-D: small textures)
    M.lds_floats = 0;
    for (int l = 0; l < M.levels; l++) {
        const bool small = M.width[l] <= TEX_LDS_MAX_WIDTH && (MODE != 2 || height <= 6 * TEX_LDS_MAX_WIDTH);
        M.lds_off[l] = small ? M.lds_floats : -1;
        if (small) M.lds_floats += (MODE == 2 ? M.width[l] * height : 6 * M.width[l] * M.width[l]) * C;
    }
    // the table has to fit behind the private copies: give up the widest private level(s) if it does not (they then go
    // through the table like the large ones)
    while ((size_t)(M.lds_floats + COMB_HT * (1 + C)) * sizeof(float) > 159 * 1024) {
        int widest = -1;
        for (int l = 0; l < M.levels; l++)
            if (M.lds_off[l] >= 0 && (widest < 0 || M.width[l] > M.width[widest])) widest = l;
        if (widest < 0) return GS2M_ERR_UNSUPPORTED;
        M.lds_off[widest] = -1;
        M.lds_floats = 0;
        for (int l = 0; l < M.levels; l++)
            if (M.lds_off[l] >= 0) { M.lds_off[l] = M.lds_floats; M.lds_floats += (MODE == 2 ? M.width[l] * height : 6 * M.width[l] * M.width[l]) * C; }
    }
    const size_t lds = (size_t)(M.lds_floats + COMB_HT * (1 + C)) * sizeof(float);
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&texture_bwd_kernel<C, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 1024) != hipSuccess)
            return GS2M_ERR_HIP;
        attr_set = true;
    }
    int blocks = (n + TEX_BWD_THREADS - 1) / TEX_BWD_THREADS;
#ifndef GS2M_TEX_BWD_BLOCKS
#define GS2M_TEX_BWD_BLOCKS 256
#endif
    if (blocks > GS2M_TEX_BWD_BLOCKS) blocks = GS2M_TEX_BWD_BLOCKS;  // one per CU: the flush costs (workgroups x private texels) atomics
    texture_bwd_kernel<C, MODE><<<blocks, TEX_BWD_THREADS, lds, s>>>(n, img_w, M, height, uv, bias, dy);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

template <int MODE>
int launch_bwd(int C, int n, int img_w, MipStack& M, int height, const float* uv, const float* bias, const float* dy, hipStream_t s) {
    switch (C) {
        case 1: return launch_bwd_c<1, MODE>(n, img_w, M, height, uv, bias, dy, s);
        case 2: return launch_bwd_c<2, MODE>(n, img_w, M, height, uv, bias, dy, s);
        case 3: return launch_bwd_c<3, MODE>(n, img_w, M, height, uv, bias, dy, s);
        case 4: return launch_bwd_c<4, MODE>(n, img_w, M, height, uv, bias, dy, s);
        default: return GS2M_ERR_UNSUPPORTED;
    }
}


// ---------------------------------------------------------------- fused deferred shading (pbr/shade.py:130-213)
// pbr_shading as ONE kernel each way: reflection vector, the three lookups (irradiance by the normal, environment BRDF by
// (N.V, roughness), prefiltered radiance by the reflection vector at the roughness' level) and the split-sum combination
//   rgb = clamp(E(n) albedo + L(r) (F0 A + B), 0, 1),   F0 = 0.04 (1 - m) + albedo m.
// The PyTorch formulation is ~30 launches forward and ~50 backward over (H, W, 3) tensors with the lookups in between.
struct ShadeIn {
    const float* normals; const float* view_dirs; const float* albedo; const float* roughness; const float* metallic;  // metallic may be NULL
    const float* lut; int lut_w, lut_h;
    const float* diffuse; float* d_diffuse; int diffuse_w, diffuse_lds;  // lds: float offset of the private copy or -1
    float min_r, max_r;
};

struct ShadePixel {  // everything both directions need for one pixel
    float n[3], a[3], m, fgA, fgB, E[3], L[3], fl;
    int l0, l1;
    bool two;
    Footprint Fd, F0, F1;
};

__device__ __forceinline__ ShadePixel shade_eval(const ShadeIn& P, const MipStack& M, size_t i) {
    ShadePixel s;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { s.n[c] = P.normals[3 * i + c]; v[c] = P.view_dirs[3 * i + c]; s.a[c] = P.albedo[3 * i + c]; }
    const float r = P.roughness[i];
    s.m = P.metallic != nullptr ? P.metallic[i] : 0.f;
    const float ndv = s.n[0] * v[0] + s.n[1] * v[1] + s.n[2] * v[2];
    const float k = 2.0f * fmaxf(ndv, 0.f);
    const float rx = k * s.n[0] - v[0], ry = k * s.n[1] - v[1], rz = k * s.n[2] - v[2];
    // CubemapLight.get_mip (pbr/light.py:72-84), then the clamp of the lookup (textureCUDA.cu:575-589)
    const int nl = M.levels;
    float fl = r < P.max_r ? (fminf(fmaxf(r, P.min_r), P.max_r) - P.min_r) / (P.max_r - P.min_r) * (float)(nl - 2)
                           : (fminf(fmaxf(r, P.max_r), 1.0f) - P.max_r) / (1.0f - P.max_r) + (float)(nl - 2);
    fl = fminf(fmaxf(fl, 0.f), (float)(nl - 1));
    s.l0 = (int)floorf(fl); s.l1 = s.l0; s.two = fl > 0.f;
    if (s.two) { s.l1 = min(s.l0 + 1, nl - 1); fl -= (float)s.l0; }
    s.fl = fl;
    s.Fd = cube_footprint(s.n[0], s.n[1], s.n[2], P.diffuse_w);
    sample<3>(P.diffuse, s.Fd, s.E);
    float fg[2];
    sample<2>(P.lut, clamp2d_footprint(fminf(fmaxf(ndv, 1e-4f), 1.0f), r, P.lut_w, P.lut_h), fg);
    s.fgA = fg[0]; s.fgB = fg[1];
    s.F0 = cube_footprint(rx, ry, rz, M.width[s.l0]);
    sample<3>(M.tex[s.l0], s.F0, s.L);
    if (s.two) {
        s.F1 = cube_footprint(rx, ry, rz, M.width[s.l1]);
        float L1[3];
        sample<3>(M.tex[s.l1], s.F1, L1);
#pragma unroll
        for (int c = 0; c < 3; c++) s.L[c] = lerpf(s.L[c], L1[c], fl);
    } else {
        s.F1 = s.F0;
    }
    return s;
}

__global__ void __launch_bounds__(256) shade_fwd_kernel(int n, ShadeIn P, MipStack M, float* __restrict__ rgb, float* __restrict__ diffuse_rgb,
                                                        float* __restrict__ specular_rgb, float* __restrict__ diffuse_light) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ShadePixel s = shade_eval(P, M, (size_t)i);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float F0 = P.metallic != nullptr ? (1.0f - s.m) * 0.04f + s.a[c] * s.m : 0.04f;
        const float d = s.E[c] * s.a[c], sp = s.L[c] * (F0 * s.fgA + s.fgB);
        rgb[3 * (size_t)i + c] = fminf(fmaxf(d + sp, 0.f), 1.f);
        if (diffuse_rgb != nullptr) diffuse_rgb[3 * (size_t)i + c] = d;
        if (specular_rgb != nullptr) specular_rgb[3 * (size_t)i + c] = sp;
        if (diffuse_light != nullptr) diffuse_light[3 * (size_t)i + c] = s.E[c];
    }
}

// The large specular levels cannot be privatised (19 MB), and one global atomic per (pixel, texel, channel) is what the
// kernel then spends its time on (0.27 ms without them, 0.83 ms with).  A workgroup therefore works on 32 x 32 PIXEL TILES
// (image width given) and puts a small combining table in LDS in front of the atomics: key = (level, texel), open
// addressing, 8 probes, then straight to memory; the table is drained after every tile.  A tile of a smooth image touches a
// few hundred distinct texels with its ~8000 contributions.
__global__ void __launch_bounds__(TEX_BWD_THREADS) shade_bwd_kernel(int n, int img_w, ShadeIn P, MipStack M, const float* __restrict__ d_rgb,
                                                                    float* __restrict__ d_albedo, float* __restrict__ d_metallic) {
    extern __shared__ float s_acc[];
    const int lds_total = M.lds_floats + (P.diffuse_lds >= 0 ? 6 * P.diffuse_w * P.diffuse_w * 3 : 0);
    CombineTable<3> T = {reinterpret_cast<int*>(s_acc + lds_total), s_acc + lds_total + COMB_HT};
    for (int k = threadIdx.x; k < lds_total; k += TEX_BWD_THREADS) s_acc[k] = 0.f;
    T.clear(threadIdx.x, TEX_BWD_THREADS);
    __syncthreads();
    auto add = [&](bool on, int key_hi, int loff, float* gptr, const Footprint& F, const float (&g)[3], float scale) {
        float wt[4];
        footprint_weights(F, wt);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool live = on && F.t[k] >= 0;
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = live ? wt[k] * scale * g[c] : 0.f;
            if (run_merge<3>(live ? (key_hi << 24) | F.t[k] : -1, v)) {
                if (loff >= 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++) atomicAdd(&s_acc[loff + F.t[k] * 3 + c], v[c]);
                } else if (key_hi < M.levels) {
                    T.add(key_hi, F.t[k], gptr, v);
                } else {  // (an irradiance map too large for LDS: not a level of the stack)
#pragma unroll
                    for (int c = 0; c < 3; c++) unsafeAtomicAdd(&gptr[(size_t)F.t[k] * 3 + c], v[c]);
                }
            }
        }
    };
    const int ntiles = tile_count(n, img_w);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        bool valid;
        const int i = tile_pixel(tile, threadIdx.x, n, img_w, valid);
        const size_t ii = valid ? (size_t)i : 0;
        const ShadePixel s = shade_eval(P, M, ii);
        float g[3], gE[3], gL[3], dm = 0.f;
        bool any = false;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float F0 = P.metallic != nullptr ? (1.0f - s.m) * 0.04f + s.a[c] * s.m : 0.04f;
            const float refl = F0 * s.fgA + s.fgB;
            const float raw = s.E[c] * s.a[c] + s.L[c] * refl;
            g[c] = (valid && raw >= 0.f && raw <= 1.f) ? d_rgb[3 * ii + c] : 0.f;  // clamp passes the gradient on [0, 1]
            any |= g[c] != 0.f;
            gE[c] = g[c] * s.a[c];
            gL[c] = g[c] * refl;
            const float dF0 = g[c] * s.L[c] * s.fgA;                                   // d/dF0
            if (valid) d_albedo[3 * ii + c] = g[c] * s.E[c] + (P.metallic != nullptr ? dF0 * s.m : 0.f);
            dm += dF0 * (s.a[c] - 0.04f);
        }
        if (valid && d_metallic != nullptr) d_metallic[ii] = dm;
        add(any, 15, P.diffuse_lds, P.d_diffuse, s.Fd, gE, 1.f);
        add(any, s.l0, M.lds_off[any ? s.l0 : 0], M.grad[any ? s.l0 : 0], s.F0, gL, s.two ? 1.f - s.fl : 1.f);
        add(any && s.two, s.l1, M.lds_off[any ? s.l1 : 0], M.grad[any ? s.l1 : 0], s.F1, gL, s.fl);
        T.drain(threadIdx.x, TEX_BWD_THREADS, [&](int level) { return M.grad[level]; });
    }
    __syncthreads();
    auto flush = [&](int loff, float* gptr, int cnt) {
        const int start = (int)(((long long)blockIdx.x * cnt) / gridDim.x);
        for (int k0 = threadIdx.x; k0 < cnt; k0 += TEX_BWD_THREADS) {
            int k = k0 + start;
            if (k >= cnt) k -= cnt;
            const float v = s_acc[loff + k];
            if (v != 0.f) unsafeAtomicAdd(&gptr[k], v);
        }
    };
    for (int l = 0; l < M.levels; l++)
        if (M.lds_off[l] >= 0) flush(M.lds_off[l], M.grad[l], 6 * M.width[l] * M.width[l] * 3);
    if (P.diffuse_lds >= 0) flush(P.diffuse_lds, P.d_diffuse, 6 * P.diffuse_w * P.diffuse_w * 3);
}

int fill_stack(MipStack& M, int levels, const float* const* tex, float* const* grad, const int* width, bool bwd) {
    if (levels < 1 || levels > GS2M_TEX_MAX_LEVELS || !width || (bwd ? !grad : !tex)) return GS2M_ERR_INVALID_ARG;
    M.levels = levels;
    for (int l = 0; l < levels; l++) {
        if (width[l] < 1 || (bwd ? !grad[l] : !tex[l])) return GS2M_ERR_INVALID_ARG;
        M.tex[l] = tex ? tex[l] : nullptr;
        M.grad[l] = grad ? grad[l] : nullptr;
        M.width[l] = width[l];
    }
    return GS2M_OK;
}

}  // namespace

extern "C" {

int gs2m_texture_cube_forward(int n, int channels, int levels, const float* const* tex, const int* width, const float* dirs,
                              const float* mip_level_bias, float* out, void* stream) {
    if (n == 0) return GS2M_OK;
    if (n < 0 || !dirs || !out) return GS2M_ERR_INVALID_ARG;
    MipStack M;
    const int rc = fill_stack(M, levels, tex, nullptr, width, false);
    if (rc != GS2M_OK) return rc;
    if (mip_level_bias) return launch_fwd<1>(channels, n, M, 0, dirs, mip_level_bias, out, (hipStream_t)stream);
    return launch_fwd<0>(channels, n, M, 0, dirs, nullptr, out, (hipStream_t)stream);
}

int gs2m_texture_cube_backward(int n, int channels, int levels, float* const* grad_tex, const int* width, const float* dirs,
                               const float* mip_level_bias, const float* dL_dout, int image_width, void* stream) {
    if (n == 0) return GS2M_OK;
    if (n < 0 || !dirs || !dL_dout || image_width < 0 || (image_width > 0 && n % image_width != 0)) return GS2M_ERR_INVALID_ARG;
    MipStack M;
    const int rc = fill_stack(M, levels, nullptr, grad_tex, width, true);
    if (rc != GS2M_OK) return rc;
    if (mip_level_bias) return launch_bwd<1>(channels, n, image_width, M, 0, dirs, mip_level_bias, dL_dout, (hipStream_t)stream);
    return launch_bwd<0>(channels, n, image_width, M, 0, dirs, nullptr, dL_dout, (hipStream_t)stream);
}

int gs2m_texture_2d_clamp_forward(int n, int channels, int width, int height, const float* tex, const float* uv, float* out,
                                  void* stream) {
    if (n == 0) return GS2M_OK;
    if (n < 0 || width < 1 || height < 1 || !tex || !uv || !out) return GS2M_ERR_INVALID_ARG;
    MipStack M;
    M.levels = 1; M.tex[0] = tex; M.grad[0] = nullptr; M.width[0] = width;
    return launch_fwd<2>(channels, n, M, height, uv, nullptr, out, (hipStream_t)stream);
}

int gs2m_texture_2d_clamp_backward(int n, int channels, int width, int height, float* grad_tex, const float* uv,
                                   const float* dL_dout, void* stream) {
    if (n == 0) return GS2M_OK;
    if (n < 0 || width < 1 || height < 1 || !grad_tex || !uv || !dL_dout) return GS2M_ERR_INVALID_ARG;
    MipStack M;
    M.levels = 1; M.tex[0] = nullptr; M.grad[0] = grad_tex; M.width[0] = width;
    return launch_bwd<2>(channels, n, 0, M, height, uv, nullptr, dL_dout, (hipStream_t)stream);
}

int gs2m_pbr_shade_forward(int n, const float* normals, const float* view_dirs, const float* albedo, const float* roughness,
                           const float* metallic, const float* brdf_lut, int lut_width, int lut_height, const float* diffuse,
                           int diffuse_width, int levels, const float* const* specular, const int* width, float min_roughness,
                           float max_roughness, float* render_rgb, float* diffuse_rgb, float* specular_rgb, float* diffuse_light,
                           void* stream) {
    if (n == 0) return GS2M_OK;
    if (n < 0 || !normals || !view_dirs || !albedo || !roughness || !brdf_lut || !diffuse || !render_rgb || lut_width < 1 ||
        lut_height < 1 || diffuse_width < 1 || levels < 2)
        return GS2M_ERR_INVALID_ARG;
    MipStack M;
    const int rc = fill_stack(M, levels, specular, nullptr, width, false);
    if (rc != GS2M_OK) return rc;
    ShadeIn P = {normals, view_dirs, albedo, roughness, metallic, brdf_lut, lut_width, lut_height, diffuse, nullptr, diffuse_width, -1,
                 min_roughness, max_roughness};
    shade_fwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, P, M, render_rgb, diffuse_rgb, specular_rgb, diffuse_light);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

int gs2m_pbr_shade_backward(int n, const float* normals, const float* view_dirs, const float* albedo, const float* roughness,
                            const float* metallic, const float* brdf_lut, int lut_width, int lut_height, const float* diffuse,
                            int diffuse_width, int levels, const float* const* specular, const int* width, float min_roughness,
                            float max_roughness, const float* dL_drender_rgb, float* dL_dalbedo, float* dL_dmetallic,
                            float* dL_ddiffuse, float* const* dL_dspecular, int image_width, void* stream) {
    if (n == 0) return GS2M_OK;
    if (image_width < 0 || (image_width > 0 && n % image_width != 0)) return GS2M_ERR_INVALID_ARG;
    if (n < 0 || !normals || !view_dirs || !albedo || !roughness || !brdf_lut || !diffuse || !dL_drender_rgb || !dL_dalbedo ||
        !dL_ddiffuse || !dL_dspecular || lut_width < 1 || lut_height < 1 || diffuse_width < 1 || levels < 2 || (dL_dmetallic && !metallic))
        return GS2M_ERR_INVALID_ARG;
    MipStack M;
    int rc = fill_stack(M, levels, specular, nullptr, width, false);
    if (rc != GS2M_OK) return rc;
    M.lds_floats = 0;
    for (int l = 0; l < levels; l++) {
        if (!dL_dspecular[l]) return GS2M_ERR_INVALID_ARG;
        M.grad[l] = dL_dspecular[l];
        const bool small = width[l] <= TEX_LDS_MAX_WIDTH;
        M.lds_off[l] = small ? M.lds_floats : -1;
        if (small) M.lds_floats += 6 * width[l] * width[l] * 3;
    }
    ShadeIn P = {normals, view_dirs, albedo, roughness, metallic, brdf_lut, lut_width, lut_height, diffuse, dL_ddiffuse, diffuse_width, -1,
                 min_roughness, max_roughness};
    int lds_floats = M.lds_floats;
    if (diffuse_width <= TEX_LDS_MAX_WIDTH) { P.diffuse_lds = lds_floats; lds_floats += 6 * diffuse_width * diffuse_width * 3; }
    lds_floats += COMB_HT * 4;  // the combining table behind the private copies
    if ((size_t)lds_floats * sizeof(float) > 159 * 1024) return GS2M_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&shade_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
            return GS2M_ERR_HIP;
        attr_set = true;
    }
    int blocks = (n + TEX_BWD_THREADS - 1) / TEX_BWD_THREADS;
    if (blocks > 256) blocks = 256;
    shade_bwd_kernel<<<blocks, TEX_BWD_THREADS, (size_t)lds_floats * sizeof(float), (hipStream_t)stream>>>(n, image_width, P, M, dL_drender_rgb, dL_dalbedo,
                                                                                                        dL_dmetallic);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

}  // extern "C"
