/*
 * gs2m_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement (plain C, fp32 arithmetic kept in the reference's expression
 * order, compiled with -ffp-contract=off) of the GS-2M differentiable Gaussian
 * rasterizer and of simple-knn's distCUDA2.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY STATUS: PINNED for the rasterizer since round 3 -- tests/test_reference_gpu.py
 * holds this file to oracle/_ref/libgs2m_ref.so, the reference's own kernels passed
 * through the image's hipify-perl and built for gfx950 (oracle/ref_build/Makefile):
 * integer artefacts and the per-Gaussian forward bit for bit, images and gradients up
 * to the summation order (DESIGN.md "Oracle").  The reference ships no tests, golden
 * vectors or CPU implementation for this path; distCUDA2 below is the exhaustive
 * 3-nearest-neighbour definition (exact by construction, not pinned to a build of
 * simple-knn).  This file follows the reference sources line by line (citations below, paths relative to
 * /root/reference/submodules/diff-gaussian-rasterization unless stated):
 *
 *   CR = cuda_rasterizer/
 *   CR/auxiliary.h:40-162        ndc2Pix, getRect, transformPoint*, dnormvdv, in_frustum
 *   CR/forward.cu:20-67          computeColorFromSH (fwd)
 *   CR/forward.cu:70-104         computeCov2D
 *   CR/forward.cu:109-142        computeCov3D
 *   CR/forward.cu:145-241        preprocessCUDA (fwd)
 *   CR/forward.cu:246-372        renderCUDA (fwd)
 *   CR/rasterizer_impl.cu:31-44  getHigherMsb
 *   CR/rasterizer_impl.cu:63-103 duplicateWithKeys
 *   CR/rasterizer_impl.cu:108-129 identifyTileRanges
 *   CR/rasterizer_impl.cu:185-330 Rasterizer::forward orchestration
 *   CR/backward.cu:23-148        computeColorFromSH (bwd)
 *   CR/backward.cu:153-281       computeCov2DCUDA
 *   CR/backward.cu:285-347       computeCov3D (bwd)
 *   CR/backward.cu:352-410       preprocessCUDA (bwd)
 *   CR/backward.cu:413-598       renderCUDA (bwd)
 *   rasterize_points.cu:63-66,150-159  zero-initialised outputs
 *   submodules/simple-knn/simple_knn.cu:44-204  distCUDA2
 *
 * GLM semantics (vendored third_party/glm): mat3(a..i) fills COLUMNS, M[i][j] is
 * column i row j, operator* is the usual matrix product with the k-sum evaluated
 * left to right (glm/detail/type_mat3x3.inl:486-519).  The m3 helpers below
 * restate exactly that so the formulas can be followed index by index.
 *
 * Float atomics in the reference accumulate per-Gaussian gradients in an
 * unspecified order; this oracle accumulates those sums in double and rounds
 * once, i.e. it is the order-independent value the reference approximates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3
#define NUM_FEATURES 10

/* CR/auxiliary.h:20-38 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                              0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                              -0.5900435899266435f};

typedef struct { float x, y, z; } f3;
typedef struct { float c[3][3]; } m3; /* c[col][row], GLM layout */

static m3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    m3 m;
    m.c[0][0] = a; m.c[0][1] = b; m.c[0][2] = c;
    m.c[1][0] = d; m.c[1][1] = e; m.c[1][2] = f;
    m.c[2][0] = g; m.c[2][1] = h; m.c[2][2] = i;
    return m;
}
static m3 m3_mul(m3 A, m3 B) { /* type_mat3x3.inl:486-519 */
    m3 R;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++)
            R.c[col][row] = A.c[0][row] * B.c[col][0] + A.c[1][row] * B.c[col][1] + A.c[2][row] * B.c[col][2];
    return R;
}
static m3 m3_T(m3 A) {
    m3 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.c[i][j] = A.c[j][i];
    return R;
}
static m3 m3_scale(float s, m3 A) {
    m3 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.c[i][j] = A.c[i][j] * s;
    return R;
}
static float dot3(const float* a, const float* b) { /* glm::dot: tmp = a*b; tmp.x+tmp.y+tmp.z */
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* CR/auxiliary.h:40-42: evaluated in double, narrowed on return */
static float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* CR/auxiliary.h:44-53 */
static void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t* rmin, uint32_t* rmax) {
    rmin[0] = (uint32_t)imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = (uint32_t)imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = (uint32_t)imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = (uint32_t)imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* CR/auxiliary.h:67-84 */
static f3 transformPoint4x3(f3 p, const float* m) {
    f3 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
    return t;
}
static void transformPoint4x4(f3 p, const float* m, float out[4]) {
    out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
/* CR/auxiliary.h:95-102 */
static f3 transformVec4x3Transpose(f3 p, const float* m) {
    f3 t = {m[0] * p.x + m[1] * p.y + m[2] * p.z,
            m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
    return t;
}
/* CR/auxiliary.h:111-120 */
static f3 dnormvdv3(f3 v, f3 dv) {
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    f3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* CR/rasterizer_impl.cu:31-44 */
static uint32_t getHigherMsb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}
uint32_t gs2m_oracle_higher_msb(uint32_t n) { return getHigherMsb(n); }

/* CR/forward.cu:20-67 */
static void sh_to_rgb_fwd(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                          const float* shs, uint8_t* clamped, float out[3]) {
    float dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dot3(dir, dir));
    dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result +
                    SH_C2[0] * xy * SH(4) +
                    SH_C2[1] * yz * SH(5) +
                    SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                    SH_C2[3] * xz * SH(7) +
                    SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result +
                        SH_C3[0] * y * (3.0f * xx - yy) * SH(9) +
                        SH_C3[1] * xy * z * SH(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                        SH_C3[5] * z * (xx - yy) * SH(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + c] = (result < 0);
        out[c] = result > 0.0f ? result : 0.0f;
    }
}

/* CR/forward.cu:109-142 */
static void computeCov3D(const float* scale, float mod, const float* rot, float* out) {
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * scale[0];
    S.c[1][1] = mod * scale[1];
    S.c[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3]; /* no normalisation, forward.cu:117 */
    m3 R = m3_cols(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 M = m3_mul(S, R);
    m3 Sigma = m3_mul(m3_T(M), M);
    out[0] = Sigma.c[0][0]; out[1] = Sigma.c[0][1]; out[2] = Sigma.c[0][2];
    out[3] = Sigma.c[1][1]; out[4] = Sigma.c[1][2]; out[5] = Sigma.c[2][2];
}

typedef struct { f3 t; float txtz, tytz; m3 J, W, T, Vrk, cov; } cov2d_tmp;

/* CR/forward.cu:70-104 (also the recomputation at CR/backward.cu:175-203) */
static void computeCov2D(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         const float* cov3D, const float* vm, cov2d_tmp* o) {
    f3 t = transformPoint4x3(mean, vm);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    o->t = t; o->txtz = txtz; o->tytz = tytz;
    o->J = m3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
                   0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
                   0, 0, 0);
    o->W = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    o->T = m3_mul(o->W, o->J);
    o->Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    o->cov = m3_mul(m3_mul(m3_T(o->T), m3_T(o->Vrk)), o->T);
}

/* ------------------------------------------------------------------------- */
typedef struct {
    int P, W, H, R, tiles_x, tiles_y, sort_bits, pad_;
    float* depths;          /* P   */
    float* means2D;         /* 2P  */
    float* cov3D;           /* 6P  */
    float* conic_opacity;   /* 4P  */
    float* rgb;             /* 3P  */
    uint8_t* clamped;       /* 3P  */
    int* radii;             /* P   */
    uint32_t* tiles_touched;/* P   */
    uint32_t* point_offsets;/* P (inclusive scan) */
    uint64_t* keys_unsorted;/* R   */
    uint32_t* vals_unsorted;/* R   */
    uint64_t* keys_sorted;  /* R   */
    uint32_t* vals_sorted;  /* R   */
    uint32_t* ranges;       /* 2*Tn */
    float* final_T;         /* W*H */
    uint32_t* n_contrib;    /* W*H */
} oracle_state;

void gs2m_oracle_free(oracle_state* s) {
    if (!s) return;
    free(s->depths); free(s->means2D); free(s->cov3D); free(s->conic_opacity); free(s->rgb);
    free(s->clamped); free(s->radii); free(s->tiles_touched); free(s->point_offsets);
    free(s->keys_unsorted); free(s->vals_unsorted); free(s->keys_sorted); free(s->vals_sorted);
    free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s);
}

/* stable LSD radix sort on key bits [0, end_bit) -- the contract of
 * cub::DeviceRadixSort::SortPairs(..., 0, 32 + bit), CR/rasterizer_impl.cu:291-296 */
static void stable_sort_pairs(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout,
                              size_t n, int end_bit) {
    uint64_t* ka = (uint64_t*)malloc((n ? n : 1) * sizeof(uint64_t));
    uint32_t* va = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint64_t* kb = (uint64_t*)malloc((n ? n : 1) * sizeof(uint64_t));
    uint32_t* vb = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    memcpy(ka, kin, n * sizeof(uint64_t));
    memcpy(va, vin, n * sizeof(uint32_t));
    size_t* cnt = (size_t*)malloc(65537 * sizeof(size_t));
    for (int shift = 0; shift < end_bit; shift += 16) {
        int bits = end_bit - shift < 16 ? end_bit - shift : 16;
        uint64_t mask = ((uint64_t)1 << bits) - 1;
        memset(cnt, 0, 65537 * sizeof(size_t));
        for (size_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (size_t d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; i++) {
            size_t p = cnt[(ka[i] >> shift) & mask]++;
            kb[p] = ka[i]; vb[p] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, n * sizeof(uint64_t));
    memcpy(vout, va, n * sizeof(uint32_t));
    free(ka); free(va); free(kb); free(vb); free(cnt);
}

/* CR/rasterizer_impl.cu:185-330 + rasterize_points.cu:63-66 (outputs zeroed here) */
oracle_state* gs2m_oracle_forward(
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* features, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered, int featureCount,
    float* out_color, int* out_radii, int* out_observe, float* out_buffer) {
    (void)prefiltered;
    const int W = width, H = height;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const size_t N = (size_t)W * H;
    const size_t Pn = P > 0 ? (size_t)P : 1;

    oracle_state* s = (oracle_state*)calloc(1, sizeof(oracle_state));
    s->P = P; s->W = W; s->H = H; s->tiles_x = gx; s->tiles_y = gy;
    s->depths = (float*)calloc(Pn, sizeof(float));
    s->means2D = (float*)calloc(2 * Pn, sizeof(float));
    s->cov3D = (float*)calloc(6 * Pn, sizeof(float));
    s->conic_opacity = (float*)calloc(4 * Pn, sizeof(float));
    s->rgb = (float*)calloc(3 * Pn, sizeof(float));
    s->clamped = (uint8_t*)calloc(3 * Pn, 1);
    s->radii = (int*)calloc(Pn, sizeof(int));
    s->tiles_touched = (uint32_t*)calloc(Pn, sizeof(uint32_t));
    s->point_offsets = (uint32_t*)calloc(Pn, sizeof(uint32_t));
    s->ranges = (uint32_t*)calloc(2 * (size_t)gx * gy, sizeof(uint32_t));
    s->final_T = (float*)calloc(N, sizeof(float));
    s->n_contrib = (uint32_t*)calloc(N, sizeof(uint32_t));

    memset(out_color, 0, NUM_CHANNELS * N * sizeof(float));
    memset(out_buffer, 0, NUM_FEATURES * N * sizeof(float));
    if (P > 0) { memset(out_radii, 0, (size_t)P * sizeof(int)); memset(out_observe, 0, (size_t)P * sizeof(int)); }

    /* ---- preprocessCUDA, CR/forward.cu:145-241 ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        out_radii[idx] = 0;
        s->tiles_touched[idx] = 0;
        f3 p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        f3 p_view = transformPoint4x3(p_orig, viewmatrix); /* in_frustum, auxiliary.h:140-162 */
        if (p_view.z <= 0.2f) continue;
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        f3 p_proj = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        const float* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * (size_t)idx;
        else {
            computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, s->cov3D + 6 * (size_t)idx);
            cov3D = s->cov3D + 6 * (size_t)idx;
        }
        cov2d_tmp c2;
        computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c2);
        float covx = c2.cov.c[0][0], covy = c2.cov.c[0][1], covz = c2.cov.c[1][1]; /* no +0.3 in forward */
        float det = (covx * covz - covy * covy);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {covz * det_inv, -covy * det_inv, covx * det_inv};
        float mid = 0.5f * (covx + covz);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
        uint32_t rmin[2], rmax[2];
        getRect(pix, piy, (int)radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) {
            float rgb[3];
            sh_to_rgb_fwd(idx, D, M, means3D, cam_pos, shs, s->clamped, rgb);
            s->rgb[3 * idx] = rgb[0]; s->rgb[3 * idx + 1] = rgb[1]; s->rgb[3 * idx + 2] = rgb[2];
        }
        s->depths[idx] = p_view.z;
        out_radii[idx] = (int)radius;
        s->means2D[2 * idx] = pix; s->means2D[2 * idx + 1] = piy;
        s->conic_opacity[4 * idx] = conic[0]; s->conic_opacity[4 * idx + 1] = conic[1];
        s->conic_opacity[4 * idx + 2] = conic[2]; s->conic_opacity[4 * idx + 3] = opacities[idx];
        s->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
    for (int i = 0; i < P; i++) s->radii[i] = out_radii[i];

    /* ---- InclusiveSum, CR/rasterizer_impl.cu:265-270 ---- */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += s->tiles_touched[i]; s->point_offsets[i] = acc; }
    const int R = P > 0 ? (int)s->point_offsets[P - 1] : 0;
    s->R = R;
    const size_t Rn = R > 0 ? (size_t)R : 1;
    s->keys_unsorted = (uint64_t*)calloc(Rn, sizeof(uint64_t));
    s->vals_unsorted = (uint32_t*)calloc(Rn, sizeof(uint32_t));
    s->keys_sorted = (uint64_t*)calloc(Rn, sizeof(uint64_t));
    s->vals_sorted = (uint32_t*)calloc(Rn, sizeof(uint32_t));

    /* ---- duplicateWithKeys, CR/rasterizer_impl.cu:63-103 ---- */
    for (int idx = 0; idx < P; idx++) {
        if (out_radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(s->means2D[2 * idx], s->means2D[2 * idx + 1], out_radii[idx], gx, gy, rmin, rmax);
            for (uint32_t y = rmin[1]; y < rmax[1]; y++)
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                    key <<= 32;
                    uint32_t dbits;
                    memcpy(&dbits, &s->depths[idx], 4);
                    key |= dbits;
                    s->keys_unsorted[off] = key;
                    s->vals_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    /* ---- SortPairs on bits [0, 32+bit), CR/rasterizer_impl.cu:288-296 ---- */
    int bit = (int)getHigherMsb((uint32_t)(gx * gy));
    s->sort_bits = 32 + bit;
    stable_sort_pairs(s->keys_unsorted, s->vals_unsorted, s->keys_sorted, s->vals_sorted, (size_t)R, 32 + bit);

    /* ---- identifyTileRanges, CR/rasterizer_impl.cu:108-129 (ranges memset to 0 at :298) ---- */
    for (int idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(s->keys_sorted[idx] >> 32);
        if (idx == 0) s->ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(s->keys_sorted[idx - 1] >> 32);
            if (currtile != prevtile) {
                s->ranges[2 * prevtile + 1] = (uint32_t)idx;
                s->ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == R - 1) s->ranges[2 * currtile + 1] = (uint32_t)R;
    }

    /* ---- renderCUDA forward, CR/forward.cu:246-372 ---- */
    const float* colors = colors_precomp ? colors_precomp : s->rgb;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            uint32_t r0 = s->ranges[2 * (ty * gx + tx)], r1 = s->ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    size_t pix_id = (size_t)W * py + px;
                    float pixfx = (float)px, pixfy = (float)py;
                    float T = 1.0f;
                    uint32_t contributor = 0, last_contributor = 0;
                    float C[NUM_CHANNELS] = {0}, F[NUM_FEATURES] = {0};
                    for (uint32_t k = r0; k < r1; k++) {
                        contributor++;
                        uint32_t id = s->vals_sorted[k];
                        float dx = s->means2D[2 * id] - pixfx, dy = s->means2D[2 * id + 1] - pixfy;
                        const float* co = s->conic_opacity + 4 * (size_t)id;
                        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        float alpha = fminf(0.99f, co[3] * expf(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break; /* done = true */
                        for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += colors[id * NUM_CHANNELS + ch] * alpha * T;
                        for (int ch = 0; ch < featureCount; ch++) F[ch] += features[id * NUM_FEATURES + ch] * alpha * T;
                        if (T > 0.5) {
#pragma omp atomic
                            out_observe[id] += 1;
                        }
                        T = test_T;
                        last_contributor = contributor;
                    }
                    s->final_T[pix_id] = T;
                    s->n_contrib[pix_id] = last_contributor;
                    for (int ch = 0; ch < NUM_CHANNELS; ch++) out_color[ch * N + pix_id] = C[ch] + T * background[ch];
                    for (int ch = 0; ch < featureCount; ch++) out_buffer[ch * N + pix_id] = F[ch];
                }
        }
    return s;
}

/* ------------------------------------------------------------------------- */
/* CR/backward.cu:23-148 */
static void sh_to_rgb_bwd(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                          const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                          float* dL_dmeans, float* dL_dshs) {
    float dir_orig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dot3(dir_orig, dir_orig));
    float dir[3] = {dir_orig[0] / len, dir_orig[1] / len, dir_orig[2] / len};
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3] = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    float x = dir[0], y = dir[1], z = dir[2];
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SH(k, c) sh[(k) * 3 + (c)]
#define DSH(k, v) do { for (int c_ = 0; c_ < 3; c_++) dL_dsh[(k) * 3 + c_] = (v) * dL_dRGB[c_]; } while (0)
    DSH(0, SH_C0);
    if (deg > 0) {
        float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
        DSH(1, dRGBdsh1); DSH(2, dRGBdsh2); DSH(3, dRGBdsh3);
        for (int c = 0; c < 3; c++) {
            dRGBdx[c] = -SH_C1 * SH(3, c);
            dRGBdy[c] = -SH_C1 * SH(1, c);
            dRGBdz[c] = SH_C1 * SH(2, c);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            DSH(4, SH_C2[0] * xy);
            DSH(5, SH_C2[1] * yz);
            DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
            DSH(7, SH_C2[3] * xz);
            DSH(8, SH_C2[4] * (xx - yy));
            for (int c = 0; c < 3; c++) {
                dRGBdx[c] += SH_C2[0] * y * SH(4, c) + SH_C2[2] * 2.f * -x * SH(6, c) + SH_C2[3] * z * SH(7, c) + SH_C2[4] * 2.f * x * SH(8, c);
                dRGBdy[c] += SH_C2[0] * x * SH(4, c) + SH_C2[1] * z * SH(5, c) + SH_C2[2] * 2.f * -y * SH(6, c) + SH_C2[4] * 2.f * -y * SH(8, c);
                dRGBdz[c] += SH_C2[1] * y * SH(5, c) + SH_C2[2] * 2.f * 2.f * z * SH(6, c) + SH_C2[3] * x * SH(7, c);
            }
            if (deg > 2) {
                DSH(9, SH_C3[0] * y * (3.f * xx - yy));
                DSH(10, SH_C3[1] * xy * z);
                DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
                DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
                DSH(14, SH_C3[5] * z * (xx - yy));
                DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
                for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (
                        SH_C3[0] * SH(9, c) * 3.f * 2.f * xy +
                        SH_C3[1] * SH(10, c) * yz +
                        SH_C3[2] * SH(11, c) * -2.f * xy +
                        SH_C3[3] * SH(12, c) * -3.f * 2.f * xz +
                        SH_C3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) +
                        SH_C3[5] * SH(14, c) * 2.f * xz +
                        SH_C3[6] * SH(15, c) * 3.f * (xx - yy));
                    dRGBdy[c] += (
                        SH_C3[0] * SH(9, c) * 3.f * (xx - yy) +
                        SH_C3[1] * SH(10, c) * xz +
                        SH_C3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) +
                        SH_C3[3] * SH(12, c) * -3.f * 2.f * yz +
                        SH_C3[4] * SH(13, c) * -2.f * xy +
                        SH_C3[5] * SH(14, c) * -2.f * yz +
                        SH_C3[6] * SH(15, c) * -3.f * 2.f * xy);
                    dRGBdz[c] += (
                        SH_C3[1] * SH(10, c) * xy +
                        SH_C3[2] * SH(11, c) * 4.f * 2.f * yz +
                        SH_C3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) +
                        SH_C3[4] * SH(13, c) * 4.f * 2.f * xz +
                        SH_C3[5] * SH(14, c) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    f3 dL_ddir = {dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB)};
    f3 dorig = {dir_orig[0], dir_orig[1], dir_orig[2]};
    f3 dL_dmean = dnormvdv3(dorig, dL_ddir);
    dL_dmeans[3 * idx] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* CR/backward.cu:285-347 */
static void computeCov3D_bwd(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                             float* dL_dscales, float* dL_drots) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S.c[0][0] = s[0]; S.c[1][1] = s[1]; S.c[2][2] = s[2];
    m3 M = m3_mul(S, R);
    const float* d = dL_dcov3Ds + 6 * (size_t)idx;
    m3 dL_dSigma = m3_cols(d[0], 0.5f * d[1], 0.5f * d[2],
                           0.5f * d[1], d[3], 0.5f * d[4],
                           0.5f * d[2], 0.5f * d[4], d[5]);
    m3 dL_dM = m3_mul(m3_scale(2.0f, M), dL_dSigma); /* 2.0f * M * dL_dSigma: (2.0f*M) first, left to right */
    m3 Rt = m3_T(R);
    m3 dL_dMt = m3_T(dL_dM);
    dL_dscales[3 * idx + 0] = dot3(Rt.c[0], dL_dMt.c[0]);
    dL_dscales[3 * idx + 1] = dot3(Rt.c[1], dL_dMt.c[1]);
    dL_dscales[3 * idx + 2] = dot3(Rt.c[2], dL_dMt.c[2]);
    for (int j = 0; j < 3; j++) { dL_dMt.c[0][j] *= s[0]; dL_dMt.c[1][j] *= s[1]; dL_dMt.c[2][j] *= s[2]; }
#define D(i, j) dL_dMt.c[i][j]
    float qx = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
    float qy = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
    float qz = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
    float qw = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
#undef D
    dL_drots[4 * idx + 0] = qx; dL_drots[4 * idx + 1] = qy; dL_drots[4 * idx + 2] = qz; dL_drots[4 * idx + 3] = qw;
}

/* CR/rasterizer_impl.cu:334-438.  All output arrays are zeroed here, as
 * rasterize_points.cu:150-159 does with torch::zeros. */
/* The per-Gaussian half of the backward: computeCov2DCUDA (CR/backward.cu:153-281) and preprocessCUDA
 * (CR/backward.cu:352-410, with computeColorFromSH :23-148 and computeCov3D :285-347) from the per-Gaussian sums the
 * blend backward produced (dL_dmeans2D, dL_dconics, dL_dcolors).  Shared by gs2m_oracle_backward and by
 * gs2m_oracle_backward_pergaussian, which lets a test feed it ANOTHER implementation's sums: the chain downstream of
 * the sums amplifies their last-bit differences for ill-conditioned covariances, so the two halves are checked
 * separately (tests/helpers.py). */
static void pergaussian_bwd(
    const oracle_state* s, int P, int D, int M, int width, int height,
    const float* means3D, const float* shs, const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    const float* dL_dmeans2D, const float* dL_dconics, const float* dL_dcolors,
    float* dL_dmeans3D, float* dL_dcov3D, float* dL_dshs, float* dL_dscales, float* dL_drots) {
    const float h_x = width / (2.0f * tan_fovx), h_y = height / (2.0f * tan_fovy);
    const float* cov3Ds = cov3D_precomp ? cov3D_precomp : s->cov3D;

    /* ---- computeCov2DCUDA, CR/backward.cu:153-281 ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* cov3D = cov3Ds + 6 * (size_t)idx;
        f3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        float dcx = dL_dconics[4 * idx], dcy = dL_dconics[4 * idx + 1], dcz = dL_dconics[4 * idx + 3];
        cov2d_tmp c2;
        computeCov2D(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c2);
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float x_grad_mul = c2.txtz < -limx || c2.txtz > limx ? 0 : 1;
        const float y_grad_mul = c2.tytz < -limy || c2.tytz > limy ? 0 : 1;
        f3 t = c2.t;
        m3 T = c2.T, W_ = c2.W, Vrk = c2.Vrk;
        float a = c2.cov.c[0][0] += 0.3f; /* backward-only low-pass, backward.cu:205-207 */
        float b = c2.cov.c[0][1];
        float c = c2.cov.c[1][1] += 0.3f;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov3D + 6 * (size_t)idx;
#define TT(i, j) T.c[i][j]
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
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
#define VV(i, j) Vrk.c[i][j]
        float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                        (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
        float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                        (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
        float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                        (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
        float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                        (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
        float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                        (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
        float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                        (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef VV
#undef TT
#define WW(i, j) W_.c[i][j]
        float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
        float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
        float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
        float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef WW
        float tz = 1.f / t.z;
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
        f3 dt = {dL_dtx, dL_dty, dL_dtz};
        f3 dL_dmean = transformVec4x3Transpose(dt, viewmatrix);
        dL_dmeans3D[3 * idx] = dL_dmean.x; dL_dmeans3D[3 * idx + 1] = dL_dmean.y; dL_dmeans3D[3 * idx + 2] = dL_dmean.z;
    }

    /* ---- preprocessCUDA backward, CR/backward.cu:352-410 ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        f3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        const float* proj = projmatrix;
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        float g0 = dL_dmeans2D[4 * idx], g1 = dL_dmeans2D[4 * idx + 1];
        float dmx = (proj[0] * m_w - proj[3] * mul1) * g0 + (proj[1] * m_w - proj[3] * mul2) * g1;
        float dmy = (proj[4] * m_w - proj[7] * mul1) * g0 + (proj[5] * m_w - proj[7] * mul2) * g1;
        float dmz = (proj[8] * m_w - proj[11] * mul1) * g0 + (proj[9] * m_w - proj[11] * mul2) * g1;
        dL_dmeans3D[3 * idx] += dmx; dL_dmeans3D[3 * idx + 1] += dmy; dL_dmeans3D[3 * idx + 2] += dmz;
        if (shs) sh_to_rgb_bwd(idx, D, M, means3D, campos, shs, s->clamped, dL_dcolors, dL_dmeans3D, dL_dshs);
        if (scales) computeCov3D_bwd(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov3D, dL_dscales, dL_drots);
    }
}

void gs2m_oracle_backward(
    const oracle_state* s, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* features, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii, int featureCount,
    const float* grad_colors, const float* grad_buffer,
    float* dL_dmeans2D /*P,4*/, float* dL_dconics /*P,4*/, float* dL_dopacities /*P*/, float* dL_dcolors /*P,3*/,
    float* dL_dmeans3D /*P,3*/, float* dL_dcov3D /*P,6*/, float* dL_dshs /*P,M,3*/, float* dL_dscales /*P,3*/,
    float* dL_drots /*P,4*/, float* dL_dfeatures /*P,10*/) {
    const int W = width, H = height;
    const size_t N = (size_t)W * H;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    (void)focal_x; (void)focal_y;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const size_t Pn = P > 0 ? (size_t)P : 1;
    memset(dL_dmeans2D, 0, 4 * Pn * sizeof(float)); memset(dL_dconics, 0, 4 * Pn * sizeof(float));
    memset(dL_dopacities, 0, Pn * sizeof(float)); memset(dL_dcolors, 0, 3 * Pn * sizeof(float));
    memset(dL_dmeans3D, 0, 3 * Pn * sizeof(float)); memset(dL_dcov3D, 0, 6 * Pn * sizeof(float));
    if (M > 0) memset(dL_dshs, 0, 3 * (size_t)M * Pn * sizeof(float));
    memset(dL_dscales, 0, 3 * Pn * sizeof(float)); memset(dL_drots, 0, 4 * Pn * sizeof(float));
    memset(dL_dfeatures, 0, NUM_FEATURES * Pn * sizeof(float));
    if (P == 0) return;

    /* double accumulators standing in for the float atomics (see header) */
    double* a_m2d = (double*)calloc(4 * Pn, sizeof(double));
    double* a_con = (double*)calloc(4 * Pn, sizeof(double));
    double* a_opa = (double*)calloc(Pn, sizeof(double));
    double* a_col = (double*)calloc(3 * Pn, sizeof(double));
    double* a_fea = (double*)calloc(NUM_FEATURES * Pn, sizeof(double));
    const float* colors = colors_precomp ? colors_precomp : s->rgb;

    /* ---- renderCUDA backward, CR/backward.cu:413-598 ---- */
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            uint32_t r0 = s->ranges[2 * (ty * gx + tx)], r1 = s->ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    size_t pix_id = (size_t)W * py + px;
                    float pixfx = (float)px, pixfy = (float)py;
                    const float T_final = s->final_T[pix_id];
                    float T = T_final;
                    uint32_t contributor = r1 - r0;
                    const uint32_t last_contributor = s->n_contrib[pix_id];
                    float accum_rec[NUM_CHANNELS] = {0}, accum_buf[NUM_FEATURES] = {0};
                    float dL_dpixel[NUM_CHANNELS], dL_dbuffer[NUM_FEATURES] = {0};
                    for (int i = 0; i < NUM_CHANNELS; i++) dL_dpixel[i] = grad_colors[i * N + pix_id];
                    for (int i = 0; i < featureCount; i++) dL_dbuffer[i] = grad_buffer[i * N + pix_id];
                    float last_alpha = 0;
                    float last_color[NUM_CHANNELS] = {0}, last_features[NUM_FEATURES] = {0};
                    const float ddelx_dx = 0.5 * W;
                    const float ddely_dy = 0.5 * H;
                    for (uint32_t k = r1; k-- > r0;) { /* back to front */
                        contributor--;
                        if (contributor >= last_contributor) continue;
                        uint32_t id = s->vals_sorted[k];
                        float dx = s->means2D[2 * id] - pixfx, dy = s->means2D[2 * id + 1] - pixfy;
                        const float* co = s->conic_opacity + 4 * (size_t)id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        const float alpha = fminf(0.99f, co[3] * G);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / (1.f - alpha);
                        const float dchannel_dcolor = alpha * T;
                        float dL_dalpha = 0.0f;
                        for (int ch = 0; ch < NUM_CHANNELS; ch++) {
                            const float c = colors[id * NUM_CHANNELS + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                            last_color[ch] = c;
                            const float dL_dchannel = dL_dpixel[ch];
                            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                            double v = (double)(dchannel_dcolor * dL_dchannel);
#pragma omp atomic
                            a_col[id * 3 + ch] += v;
                        }
                        for (int ch = 0; ch < featureCount; ch++) {
                            const float c = features[id * NUM_FEATURES + ch];
                            accum_buf[ch] = last_alpha * last_features[ch] + (1.f - last_alpha) * accum_buf[ch];
                            last_features[ch] = c;
                            const float dL_dchannel = dL_dbuffer[ch];
                            dL_dalpha += (c - accum_buf[ch]) * dL_dchannel;
                            double v = (double)(dchannel_dcolor * dL_dchannel);
#pragma omp atomic
                            a_fea[id * NUM_FEATURES + ch] += v;
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        float bg_dot_dpixel = 0;
                        for (int i = 0; i < NUM_CHANNELS; i++) bg_dot_dpixel += background[i] * dL_dpixel[i];
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                        const float dL_dG = co[3] * dL_dalpha;
                        const float gdx = G * dx;
                        const float gdy = G * dy;
                        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const float dG_ddely = -gdy * co[2] - gdx * co[1];
                        double v0 = (double)(dL_dG * dG_ddelx * ddelx_dx);
                        double v1 = (double)(dL_dG * dG_ddely * ddely_dy);
                        double v2 = (double)fabsf(dL_dG * dG_ddelx * ddelx_dx);
                        double v3 = (double)fabsf(dL_dG * dG_ddely * ddely_dy);
                        double c0 = (double)(-0.5f * gdx * dx * dL_dG);
                        double c1 = (double)(-0.5f * gdx * dy * dL_dG);
                        double c3 = (double)(-0.5f * gdy * dy * dL_dG);
                        double op = (double)(G * dL_dalpha);
#pragma omp atomic
                        a_m2d[4 * id + 0] += v0;
#pragma omp atomic
                        a_m2d[4 * id + 1] += v1;
#pragma omp atomic
                        a_m2d[4 * id + 2] += v2;
#pragma omp atomic
                        a_m2d[4 * id + 3] += v3;
#pragma omp atomic
                        a_con[4 * id + 0] += c0;
#pragma omp atomic
                        a_con[4 * id + 1] += c1;
#pragma omp atomic
                        a_con[4 * id + 3] += c3;
#pragma omp atomic
                        a_opa[id] += op;
                    }
                }
        }
    for (size_t i = 0; i < 4 * Pn; i++) { dL_dmeans2D[i] = (float)a_m2d[i]; dL_dconics[i] = (float)a_con[i]; }
    for (size_t i = 0; i < Pn; i++) dL_dopacities[i] = (float)a_opa[i];
    for (size_t i = 0; i < 3 * Pn; i++) dL_dcolors[i] = (float)a_col[i];
    for (size_t i = 0; i < NUM_FEATURES * Pn; i++) dL_dfeatures[i] = (float)a_fea[i];
    free(a_m2d); free(a_con); free(a_opa); free(a_col); free(a_fea);

    pergaussian_bwd(s, P, D, M, width, height, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                    projmatrix, campos, tan_fovx, tan_fovy, radii, dL_dmeans2D, dL_dconics, dL_dcolors, dL_dmeans3D,
                    dL_dcov3D, dL_dshs, dL_dscales, dL_drots);
}

/* Per-Gaussian half only, from given sums (zero-fills its outputs first, as the binding does). */
void gs2m_oracle_backward_pergaussian(
    const oracle_state* s, int P, int D, int M, int width, int height,
    const float* means3D, const float* shs, const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii,
    const float* dL_dmeans2D, const float* dL_dconics, const float* dL_dcolors,
    float* dL_dmeans3D, float* dL_dcov3D, float* dL_dshs, float* dL_dscales, float* dL_drots) {
    const size_t Pn = P > 0 ? (size_t)P : 1;
    memset(dL_dmeans3D, 0, 3 * Pn * sizeof(float)); memset(dL_dcov3D, 0, 6 * Pn * sizeof(float));
    if (M > 0) memset(dL_dshs, 0, 3 * (size_t)M * Pn * sizeof(float));
    memset(dL_dscales, 0, 3 * Pn * sizeof(float)); memset(dL_drots, 0, 4 * Pn * sizeof(float));
    if (P == 0) return;
    pergaussian_bwd(s, P, D, M, width, height, means3D, shs, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                    projmatrix, campos, tan_fovx, tan_fovy, radii, dL_dmeans2D, dL_dconics, dL_dcolors, dL_dmeans3D,
                    dL_dcov3D, dL_dshs, dL_dscales, dL_drots);
}

/* CR/rasterizer_impl.cu:48-59, 132-143 */
void gs2m_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        f3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        f3 v = transformPoint4x3(p, viewmatrix);
        present[idx] = v.z <= 0.2f ? 0 : 1;
    }
}

/* ---- simple-knn distCUDA2, submodules/simple-knn/simple_knn.cu:110-167 ----
 * The reference's Morton/box search only prunes candidates that cannot enter the
 * 3-best set, so the result is the exact 3-NN mean of squared
 * distances; an exhaustive scan with the same updateKBest arithmetic gives the
 * same floats. */
static void updateKBest3(const float* ref, const float* point, float* knn) {
    float dx = point[0] - ref[0], dy = point[1] - ref[1], dz = point[2] - ref[2];
    float dist = dx * dx + dy * dy + dz * dz;
    for (int j = 0; j < 3; j++) {
        if (knn[j] > dist) { float t = knn[j]; knn[j] = dist; dist = t; }
    }
}
void gs2m_oracle_knn_dist2(int P, const float* points, float* mean_dists) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            updateKBest3(points + 3 * (size_t)i, points + 3 * (size_t)j, best);
        }
        mean_dists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}

/* Morton code of simple_knn.cu:44-58, exposed so tests can pin the HIP kernel's codes */
static uint32_t prepMorton(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
uint32_t gs2m_oracle_morton(const float* coord, const float* minn, const float* maxx) {
    uint32_t x = prepMorton((uint32_t)(((coord[0] - minn[0]) / (maxx[0] - minn[0])) * ((1 << 10) - 1)));
    uint32_t y = prepMorton((uint32_t)(((coord[1] - minn[1]) / (maxx[1] - minn[1])) * ((1 << 10) - 1)));
    uint32_t z = prepMorton((uint32_t)(((coord[2] - minn[2]) / (maxx[2] - minn[2])) * ((1 << 10) - 1)));
    return x | (y << 1) | (z << 2);
}
