// Fused SSIM map (11x11 separable Gaussian window, sigma 1.5, zero "same" padding), forward and backward, for gfx950
// (the ROCm replacement of submodules/fused-ssim, the D-SSIM term of train.py:103 and :136).
//
// Same operator as utils/loss_utils.py:30-70 `ssim` (five depthwise 11x11 conv2d + ~15 elementwise ops), same op
// boundary as the reference's extension (fused_ssim/__init__.py:8-41): forward returns the map plus the three
// partial derivatives the backward needs, backward turns dL/dmap into dL/dimg1.
//
// Layout for wave64: ONE WAVE walks down a 64-column strip.  Per image row it stages the 74 halo columns of both
// images in LDS, every lane forms the five horizontal sums (a, a^2, b, b^2, ab) for its column, and feeds them into
// a ring of 11 vertical accumulators per quantity that lives in registers (scatter form: row t adds w[k]*h into the
// output that started k rows ago) -- so the vertical pass costs no LDS traffic at all and each input row is read
// from HBM once per strip (+10 halo rows per SSIM_ROWS).  The epilogue (SSIM value and derivatives) runs on the
// row that completes.  HBM: 8 B read + 16 B written per element forward, 24 + 4 B backward; the kernels are VALU
// bound (~200 lane-ops per element), not HBM bound.
#include "common.h"
#include "../../include/gs2m_ssim.h"

namespace {

// torch.Tensor([exp(-(x - 5)^2 / (2 * 1.5^2)) for x in range(11)]) / sum, in fp32 (utils/loss_utils.py:41-43)
__device__ __constant__ const float SSIM_W[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                                  0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                                  0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                                  0.0075987582094967365f, 0.001028380123898387f};

constexpr int SSIM_ROWS = 45;   // output rows per strip at most (1080 = 24 * 45); +10 halo rows of input.  ssim_rows() below picks fewer on small frames
constexpr int SSIM_LDSW = 80;   // 64 + 10 halo columns, padded
constexpr int SSIM_PF_FWD = 3;  // input rows in flight ahead of the row being consumed: the strip walk is a chain of
constexpr int SSIM_PF_BWD = 5;  // dependent HBM round trips (~1 us each) with only ~2 waves per SIMD to hide them

template <int NQ>
struct Ring {  // NQ quantities x 11 vertical accumulators
    float a[NQ][11];
};

// ---------------------------------------------------------------- forward
__global__ void __launch_bounds__(64) ssim_fwd_kernel(int H, int W, int rows, float C1, float C2, const float* __restrict__ img1,
                                                      const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                      float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                      float* __restrict__ dm_dsigma12) {
    __shared__ float s_a[2][SSIM_LDSW], s_b[2][SSIM_LDSW];
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * rows;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float* __restrict__ A = img1 + plane;
    const float* __restrict__ Bm = img2 + plane;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = SSIM_W[k];

    const int xa = x0 - 5 + lane, xb = x0 + 59 + lane;  // main column and (lanes 0..9) the right halo column
    const bool ina = xa >= 0 && xa < W, inb = lane < 10 && xb < W;
    auto fetch = [&](int y, float& a0, float& b0, float& a1, float& b1) {
        const bool row = y >= 0 && y < H;
        const size_t o = (size_t)(row ? y : 0) * W;
        a0 = (row && ina) ? A[o + xa] : 0.f;
        b0 = (row && ina) ? Bm[o + xa] : 0.f;
        a1 = (row && inb) ? A[o + xb] : 0.f;
        b1 = (row && inb) ? Bm[o + xb] : 0.f;
    };

    Ring<5> R;
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
        for (int j = 0; j < 11; j++) R.a[q][j] = 0.f;

    const int nrows = min(rows, H - y0) + 10;  // input rows y0-5 .. y0+rows+4
    float pf[SSIM_PF_FWD][4];  // queue of fetched rows: pf[0] is the next one to consume
#pragma unroll
    for (int d = 0; d < SSIM_PF_FWD; d++) fetch(d < nrows ? y0 - 5 + d : -1, pf[d][0], pf[d][1], pf[d][2], pf[d][3]);
    const int x = x0 + lane;
    for (int t0 = 0; t0 < nrows; t0 += 11) {
#pragma unroll
        for (int u = 0; u < 11; u++) {
            const int t = t0 + u;
            if (t < nrows) {  // wave-uniform
                const int buf = u & 1;
                s_a[buf][lane] = pf[0][0]; s_b[buf][lane] = pf[0][1];
                if (lane < 10) { s_a[buf][64 + lane] = pf[0][2]; s_b[buf][64 + lane] = pf[0][3]; }
#pragma unroll
                for (int d = 0; d + 1 < SSIM_PF_FWD; d++)
#pragma unroll
                    for (int c = 0; c < 4; c++) pf[d][c] = pf[d + 1][c];
                {
                    constexpr int D = SSIM_PF_FWD - 1;
                    fetch(t + SSIM_PF_FWD < nrows ? y0 - 5 + t + SSIM_PF_FWD : -1, pf[D][0], pf[D][1], pf[D][2], pf[D][3]);
                }
                __syncthreads();
                float h[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const float a = s_a[buf][lane + k], b = s_b[buf][lane + k];
                    const float wa = w[k] * a, wb = w[k] * b;
                    h[0] += wa; h[1] = __builtin_fmaf(wa, a, h[1]);
                    h[2] += wb; h[3] = __builtin_fmaf(wb, b, h[3]);
                    h[4] = __builtin_fmaf(wa, b, h[4]);
                }
                // input row t is tap k of the output row t - k: ring slot (u - k) mod 11
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const int j = (u - k + 11) % 11;
#pragma unroll
                    for (int q = 0; q < 5; q++) R.a[q][j] = __builtin_fmaf(w[k], h[q], R.a[q][j]);
                }
                const int jo = (u + 1) % 11;  // output row t - 10 is complete
                const int yo = y0 + t - 10;
                if (t >= 10) {
                    const float mu1 = R.a[0][jo], mu2 = R.a[2][jo];
                    const float sigma1_sq = R.a[1][jo] - mu1 * mu1;
                    const float sigma2_sq = R.a[3][jo] - mu2 * mu2;
                    const float sigma12 = R.a[4][jo] - mu1 * mu2;
                    const float Cc = 2.f * (mu1 * mu2) + C1;
                    const float Dd = 2.f * sigma12 + C2;
                    const float Aa = mu1 * mu1 + mu2 * mu2 + C1;
                    const float Bb = sigma1_sq + sigma2_sq + C2;
                    const float rAB = 1.f / (Aa * Bb);
                    if (x < W) {
                        const size_t o = plane + (size_t)yo * W + x;
                        ssim_map[o] = (Cc * Dd) * rAB;
                        if (dm_dmu1 != nullptr) {
                            // d/dmu1 of (C D)/(A B) with sigma1_sq, sigma12 depending on mu1 through -mu1^2, -mu1 mu2
                            dm_dmu1[o] = (mu2 * 2.f * Dd) * rAB - (mu2 * 2.f * Cc) * rAB - (mu1 * 2.f * Cc * Dd) * rAB / Aa +
                                         (mu1 * 2.f * Cc * Dd) * rAB / Bb;
                            dm_dsigma1_sq[o] = (-Cc * Dd) * rAB / Bb;
                            dm_dsigma12[o] = (2.f * Cc) * rAB;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 5; q++) R.a[q][jo] = 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------- backward
// dL/dimg1 = conv(dL dm/dmu1) + 2 img1 conv(dL dm/dsigma1_sq) + img2 conv(dL dm/dsigma12)   (the window is symmetric)
__global__ void __launch_bounds__(64) ssim_bwd_kernel(int H, int W, int rows, const float* __restrict__ img1,
                                                      const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                      const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
                                                      const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1,
                                                      const float* __restrict__ dL_dvalue, float mul, float div) {
    __shared__ float s_x[2][3][SSIM_LDSW];
    // dL_dmap == nullptr: the map's gradient is the same at every element, dL_dvalue[0] * mul / div (the mean's backward)
    const float gu = dL_dmap == nullptr ? (dL_dvalue[0] * mul) / div : 0.f;
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * rows;
    const size_t plane = (size_t)blockIdx.z * H * W;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = SSIM_W[k];
    const int xa = x0 - 5 + lane, xb = x0 + 59 + lane;
    const bool ina = xa >= 0 && xa < W, inb = lane < 10 && xb < W;
    auto fetch = [&](int y, float (&v)[8]) {  // dL and the three derivative maps, main and halo column
        const bool row = y >= 0 && y < H;
        const size_t o = plane + (size_t)(row ? y : 0) * W;
        const bool m0 = row && ina, m1 = row && inb;
        v[0] = m0 ? (dL_dmap ? dL_dmap[o + xa] : gu) : 0.f;
        v[1] = m0 ? dm_dmu1[o + xa] : 0.f;
        v[2] = m0 ? dm_dsigma1_sq[o + xa] : 0.f;
        v[3] = m0 ? dm_dsigma12[o + xa] : 0.f;
        v[4] = m1 ? (dL_dmap ? dL_dmap[o + xb] : gu) : 0.f;
        v[5] = m1 ? dm_dmu1[o + xb] : 0.f;
        v[6] = m1 ? dm_dsigma1_sq[o + xb] : 0.f;
        v[7] = m1 ? dm_dsigma12[o + xb] : 0.f;
    };
    Ring<3> R;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int j = 0; j < 11; j++) R.a[q][j] = 0.f;
    const int nrows = min(rows, H - y0) + 10;
    float pf[SSIM_PF_BWD][8];
#pragma unroll
    for (int d = 0; d < SSIM_PF_BWD; d++) fetch(d < nrows ? y0 - 5 + d : -1, pf[d]);
    const int x = x0 + lane;
    for (int t0 = 0; t0 < nrows; t0 += 11) {
#pragma unroll
        for (int u = 0; u < 11; u++) {
            const int t = t0 + u;
            if (t < nrows) {
                const int buf = u & 1;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    s_x[buf][q][lane] = pf[0][0] * pf[0][1 + q];
                    if (lane < 10) s_x[buf][q][64 + lane] = pf[0][4] * pf[0][5 + q];
                }
#pragma unroll
                for (int d = 0; d + 1 < SSIM_PF_BWD; d++)
#pragma unroll
                    for (int c = 0; c < 8; c++) pf[d][c] = pf[d + 1][c];
                fetch(t + SSIM_PF_BWD < nrows ? y0 - 5 + t + SSIM_PF_BWD : -1, pf[SSIM_PF_BWD - 1]);
                __syncthreads();
                float h[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 11; k++)
#pragma unroll
                    for (int q = 0; q < 3; q++) h[q] = __builtin_fmaf(w[k], s_x[buf][q][lane + k], h[q]);
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const int j = (u - k + 11) % 11;
#pragma unroll
                    for (int q = 0; q < 3; q++) R.a[q][j] = __builtin_fmaf(w[k], h[q], R.a[q][j]);
                }
                const int jo = (u + 1) % 11;
                const int yo = y0 + t - 10;
                if (t >= 10 && x < W) {
                    const size_t o = plane + (size_t)yo * W + x;
                    const float a = img1[o], b = img2[o];
                    dL_dimg1[o] = R.a[0][jo] + 2.f * a * R.a[1][jo] + b * R.a[2][jo];
                }
#pragma unroll
                for (int q = 0; q < 3; q++) R.a[q][jo] = 0.f;
            }
        }
    }
}

// Rows per strip: a strip is ONE wave walking its rows as a chain of dependent memory round trips, so a frame has to be cut into
// enough strips to fill the chip (~2000 waves): 45 rows at 1080p (2160 strips), 12 at 777 x 581 (1900 strips instead of 507, whose
// 55-row chains made the kernels 55 us each on a frame a fifth of 1080p's size).  The halo (10 input rows per strip) is re-read from L2.
inline int ssim_rows(int B, int CH, int H, int W) {
    const long long cols = (W + 63) / 64;
    const long long r = (cols * H * (long long)B * CH) / 2048;
    return (int)(r < 12 ? 12 : (r > SSIM_ROWS ? SSIM_ROWS : r));
}
inline bool ssim_dims_ok(int B, int CH, int H, int W) {
    return B > 0 && CH > 0 && H > 0 && W > 0 && (long long)B * CH <= 65535 && (H + 11) / 12 <= 65535;
}

}  // namespace

extern "C" int gs2m_ssim_forward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2,
                                 float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (B == 0 || CH == 0 || H == 0 || W == 0) return GS2M_OK;
    if (B < 0 || CH < 0 || H < 0 || W < 0 || !img1 || !img2 || !ssim_map) return GS2M_ERR_INVALID_ARG;
    const bool train = dm_dmu1 || dm_dsigma1_sq || dm_dsigma12;
    if (train && !(dm_dmu1 && dm_dsigma1_sq && dm_dsigma12)) return GS2M_ERR_INVALID_ARG;
    if (!ssim_dims_ok(B, CH, H, W)) return GS2M_ERR_UNSUPPORTED;
    const int rows = ssim_rows(B, CH, H, W);
    dim3 grid((W + 63) / 64, (H + rows - 1) / rows, B * CH);
    ssim_fwd_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(H, W, rows, C1, C2, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq,
                                                         dm_dsigma12);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

extern "C" int gs2m_ssim_backward(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                                  const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                                  float* dL_dimg1, void* stream) {
    if (B == 0 || CH == 0 || H == 0 || W == 0) return GS2M_OK;
    if (B < 0 || CH < 0 || H < 0 || W < 0 || !img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return GS2M_ERR_INVALID_ARG;
    if (!ssim_dims_ok(B, CH, H, W)) return GS2M_ERR_UNSUPPORTED;
    const int rows = ssim_rows(B, CH, H, W);
    dim3 grid((W + 63) / 64, (H + rows - 1) / rows, B * CH);
    ssim_bwd_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(H, W, rows, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                                                         dL_dimg1, nullptr, 0.f, 1.f);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}

extern "C" int gs2m_ssim_backward_uniform(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dvalue,
                                          float mul, float div, const float* dm_dmu1, const float* dm_dsigma1_sq,
                                          const float* dm_dsigma12, float* dL_dimg1, void* stream) {
    if (B == 0 || CH == 0 || H == 0 || W == 0) return GS2M_OK;
    if (B < 0 || CH < 0 || H < 0 || W < 0 || !img1 || !img2 || !dL_dvalue || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1 || div == 0.f)
        return GS2M_ERR_INVALID_ARG;
    if (!ssim_dims_ok(B, CH, H, W)) return GS2M_ERR_UNSUPPORTED;
    const int rows = ssim_rows(B, CH, H, W);
    dim3 grid((W + 63) / 64, (H + rows - 1) / rows, B * CH);
    ssim_bwd_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(H, W, rows, img1, img2, nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                                                         dL_dimg1, dL_dvalue, mul, div);
    return hipGetLastError() == hipSuccess ? GS2M_OK : GS2M_ERR_HIP;
}
