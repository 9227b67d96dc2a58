// Pre-path alignment on the GPU (include/vistaf_align.h; Code/shape_ftp.py:1471-1537): BGR2GRAY, phase-correlation
// global shift, fixed-point warpAffine, ROI crop, ECC (euclidean) crop alignment.
//
// Arithmetic follows OpenCV 4.x (the test suite checks it against a CPU restatement of the same routines):
//   * BGR2GRAY on uint8: OpenCV 4.x's 15-bit fixed point, (B*3735 + G*19235 + R*9798 + 2^14) >> 15 (the same form as k_to_gray), or,
//     with vistaf_align_config.gray_coeffs = 1, OpenCV 3.x's (R*4899 + G*9617 + B*1868 + 2^13) >> 14;
//   * phaseCorrelate: sqrt-Hanning window, R2C FFTs (hipFFT), unit-magnitude cross-power spectrum, inverse FFT, arg-max in
//     fftshift order (first maximum in row-major order), 5x5 weighted centroid, shift = centre - centroid;
//   * warpAffine INTER_LINEAR: source coordinates in AB_BITS = 10 fixed point, rounded to 1/32 pixel, uint8 through the
//     15-bit integer weights, float32 with float weights; INTER_NEAREST for the ECC mask;
//   * findTransformECC (MOTION_EUCLIDEAN, gaussFiltSize = 1): per iteration ONE fused kernel warps image and gradients
//     on the fly and accumulates the 21 sums the update needs (zero-mean correlation, 3x3 Hessian, projections; the
//     means are folded in algebraically, sums in float64, per-block partials combined in a fixed order), and a one-wave
//     kernel per frame solves the 3x3 system and updates the warp.  A frame whose update fails (lambda_d <= 0, NaN)
//     returns the unaligned crop and the identity warp, exactly what the reference's `except cv2.error` branch does.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vistaf_ftp.h"
#include "../../include/vistaf_align.h"
#include "kernels.hpp"

namespace vf { int set_error(int code, const std::string &msg); }
using namespace vf;

#define ACHK(x)                                                                                              \
    do {                                                                                                     \
        hipError_t e_ = (x);                                                                                 \
        if (e_ != hipSuccess) return set_error(VISTAF_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define FCHK(x)                                                                                        \
    do {                                                                                               \
        hipfftResult r_ = (x);                                                                         \
        if (r_ != HIPFFT_SUCCESS) return set_error(VISTAF_E_HIP, std::string(#x) + ": hipfft error " + std::to_string((int)r_)); \
    } while (0)

namespace {

constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS;
constexpr int ECC_NSUM = 21, ECC_BLOCKS = 128, ECC_T = 256;

struct Aff { double m[6]; };      // source = M * (x, y, 1): the inverse map warpAffine iterates with

__device__ inline long long cvr(double v) { return __double2ll_rn(v); }          // cvRound / saturate_cast<int>(double): half to even

// fixed-point source coordinate of destination pixel (x, y): integer part and 1/32 fraction (WarpAffineInvoker)
__device__ inline void src_coord(const Aff &a, int x, int y, int round_delta, int shift, int &X, int &Y)
{
    long long ad = cvr(a.m[0] * x * AB_SCALE), bd = cvr(a.m[3] * x * AB_SCALE);
    long long X0 = cvr((a.m[1] * y + a.m[2]) * AB_SCALE) + round_delta, Y0 = cvr((a.m[4] * y + a.m[5]) * AB_SCALE) + round_delta;
    X = (int)((X0 + ad) >> shift);
    Y = (int)((Y0 + bd) >> shift);
}
__device__ inline int reflect_b(int p, int n)       // BORDER_REFLECT
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

__global__ void k_bgr2gray(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ g8, float *__restrict__ gf, size_t n, int coeffs3x)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
    int v = coeffs3x ? (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14 : (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15;
    if (g8) g8[i] = (uint8_t)v;
    if (gf) gf[i] = (float)v;
}

// img *= sqrt(hann_row * hann_col)  (cv::createHanningWindow, CV_32F)
__global__ void k_hann_mul(float *__restrict__ img, int h, int w)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    double wc = 0.5 * (1.0 - cos(2.0 * M_PI / (double)(w - 1) * x)), wr = 0.5 * (1.0 - cos(2.0 * M_PI / (double)(h - 1) * y));
    float win = sqrtf((float)(wr * wc));
    img[(size_t)y * w + x] *= win;
}

// C = F1 * conj(F2) / |F1 * conj(F2)| on the half spectrum
__global__ void k_cross_power(const float2 *__restrict__ f1, const float2 *__restrict__ f2, float2 *__restrict__ c, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 a = f1[i], b = f2[i];
    float re = a.x * b.x + a.y * b.y, im = a.y * b.x - a.x * b.y;
    float mag = sqrtf(re * re + im * im);
    c[i] = mag > 0.f ? make_float2(re / mag, im / mag) : make_float2(0.f, 0.f);
}

// arg-max of the correlation surface in fftshift order: key = (value bits, first index in shifted row-major order)
__global__ __launch_bounds__(1024) void k_peak_partial(const float *__restrict__ corr, int M, int N, unsigned long long *__restrict__ part)
{
    __shared__ unsigned long long red[16];
    const size_t total = (size_t)M * N;
    unsigned long long best = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(i / N), x = (int)(i - (size_t)y * N);
        int ys = (y + M / 2) % M, xs = (x + N / 2) % N;                    // position after cv's fftShift (even sizes)
        uint32_t sidx = (uint32_t)((size_t)ys * N + xs);
        unsigned long long key = ((unsigned long long)f2key(corr[i]) << 32) | (uint32_t)(0xffffffffu - sidx);
        best = key > best ? key : best;
    }
    best = block_max_u64(best, red);
    if (threadIdx.x == 0) part[blockIdx.x] = best;
}
// one wave: final arg-max, 5x5 weighted centroid (cv::weightedCentroid), shift = centre - centroid
__global__ void k_peak_final(const float *__restrict__ corr, int M, int N, const unsigned long long *__restrict__ part, int nparts,
                             double *__restrict__ info)
{
    unsigned long long best = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) best = part[i] > best ? part[i] : best;
    best = wave_max_u64(best);
    const uint32_t sidx = 0xffffffffu - (uint32_t)best;
    const int py = (int)(sidx / (uint32_t)N), px = (int)(sidx - (uint32_t)py * N);
    // 25 window cells on lanes 0..24
    const int l = threadIdx.x, dy = l / 5 - 2, dx = l % 5 - 2;
    const int ys = py + dy, xs = px + dx;
    double v = 0.0;
    if (l < 25 && ys >= 0 && ys < M && xs >= 0 && xs < N) {
        int y = (ys + M - M / 2) % M, x = (xs + N - N / 2) % N;            // back to unshifted storage
        v = (double)corr[(size_t)y * N + x];
    }
    double s = wave_sum(v), sx = wave_sum(v * xs), sy = wave_sum(v * ys);
    if (threadIdx.x == 0) {
        info[VISTAF_AI_SHIFT_X] = N / 2.0 - sx / s;
        info[VISTAF_AI_SHIFT_Y] = M / 2.0 - sy / s;
        info[VISTAF_AI_RESPONSE] = s / ((double)M * N);
    }
}

// dst = warpAffine(src u8, C channels) with inverse map `a`, INTER_LINEAR, BORDER_REFLECT; only the window [x1,x2) x [y1,y2) of
// the destination is produced (dst is the window, row pitch (x2 - x1) * C)
template <int C>
__global__ void k_warp_u8(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, Aff a, int h, int w, int x1, int y1, int x2, int y2)
{
    int x = x1 + blockIdx.x * blockDim.x + threadIdx.x, y = y1 + blockIdx.y;
    if (x >= x2) return;
    int X, Y;
    src_coord(a, x, y, AB_SCALE / INTER_TAB / 2, AB_BITS - INTER_BITS, X, Y);
    const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS, ax = X & (INTER_TAB - 1), ay = Y & (INTER_TAB - 1);
    const int xa = reflect_b(sx, w), xb = reflect_b(sx + 1, w), ya = reflect_b(sy, h), yb = reflect_b(sy + 1, h);
    const int w00 = (INTER_TAB - ax) * (INTER_TAB - ay) * 32, w01 = ax * (INTER_TAB - ay) * 32, w10 = (INTER_TAB - ax) * ay * 32, w11 = ax * ay * 32;
    uint8_t *o = dst + ((size_t)(y - y1) * (x2 - x1) + (x - x1)) * C;
#pragma unroll
    for (int c = 0; c < C; c++) {
        int acc = src[((size_t)ya * w + xa) * C + c] * w00 + src[((size_t)ya * w + xb) * C + c] * w01 + src[((size_t)yb * w + xa) * C + c] * w10 +
                  src[((size_t)yb * w + xb) * C + c] * w11;
        int v = (acc + (1 << 14)) >> 15;
        o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

__global__ void k_crop_bgr(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int x1, int y1, int cw)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cw) return;
    const uint8_t *s = src + ((size_t)(y1 + y) * w + x1 + x) * 3;
    uint8_t *o = dst + ((size_t)y * cw + x) * 3;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
}

__global__ void k_u8_to_unit(const uint8_t *__restrict__ s, float *__restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)s[i] / 255.0f;
}

// gradients of the ECC input image: filter2D with (-0.5, 0, 0.5), BORDER_REFLECT_101, times the mask
__global__ void k_ecc_grad(const float *__restrict__ img, const uint8_t *__restrict__ mask, float *__restrict__ gx, float *__restrict__ gy, int h, int w)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const float *I = img + b * (size_t)h * w;
    size_t p = (size_t)y * w + x;
    float m = mask[p] ? 1.f : 0.f;
    float l = I[(size_t)y * w + reflect101(x - 1, w)], r = I[(size_t)y * w + reflect101(x + 1, w)];
    float u = I[(size_t)reflect101(y - 1, h) * w + x], d = I[(size_t)reflect101(y + 1, h) * w + x];
    gx[b * (size_t)h * w + p] = (0.5f * r - 0.5f * l) * m;
    gy[b * (size_t)h * w + p] = (0.5f * d - 0.5f * u) * m;
}

struct EccState {            // per frame, device
    float warp[6];
    double rho, last_rho;
    int iter, done, failed, pad;
};

__device__ inline float bilin_f(const float *__restrict__ s, int h, int w, int sx, int sy, float w00, float w01, float w10, float w11)
{
    // BORDER_CONSTANT 0
    const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w, y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
    float p00 = (x0 && y0) ? s[(size_t)sy * w + sx] : 0.f, p01 = (x1 && y0) ? s[(size_t)sy * w + sx + 1] : 0.f;
    float p10 = (x0 && y1) ? s[(size_t)(sy + 1) * w + sx] : 0.f, p11 = (x1 && y1) ? s[(size_t)(sy + 1) * w + sx + 1] : 0.f;
    return p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
}

// One ECC iteration, part 1: warp image / gradients / mask with the current map and accumulate the 21 sums.
//   0 n | 1 S i | 2 S i^2 | 3 S t | 4 S t^2 | 5 S t*i   (inside the warped mask)
//   6..11 Hessian J_a.J_b (a <= b, all pixels) | 12..14 S J_a*i (all) | 15..17 S J_a (inside) | 18..20 S J_a*t (inside)
__global__ __launch_bounds__(ECC_T) void k_ecc_accumulate(const float *__restrict__ tpl, const float *__restrict__ img_all, const float *__restrict__ gx_all,
                                                          const float *__restrict__ gy_all, const uint8_t *__restrict__ premask,
                                                          const EccState *__restrict__ st_all, double *__restrict__ partial, int h, int w)
{
    __shared__ double red[ECC_T / 64][ECC_NSUM];
    const size_t b = blockIdx.y;
    const EccState &st = st_all[b];
    if (st.done) return;
    const size_t P = (size_t)h * w;
    const float *img = img_all + b * P, *gx = gx_all + b * P, *gy = gy_all + b * P;
    Aff a;
    for (int i = 0; i < 6; i++) a.m[i] = (double)st.warp[i];
    const float h0 = st.warp[0], h1 = st.warp[3];
    double acc[ECC_NSUM];
#pragma unroll
    for (int i = 0; i < ECC_NSUM; i++) acc[i] = 0.0;
    for (size_t p = (size_t)blockIdx.x * ECC_T + threadIdx.x; p < P; p += (size_t)gridDim.x * ECC_T) {
        const int y = (int)(p / w), x = (int)(p - (size_t)y * w);
        int X, Y;
        src_coord(a, x, y, AB_SCALE / INTER_TAB / 2, AB_BITS - INTER_BITS, X, Y);
        const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
        const float fx = (float)(X & (INTER_TAB - 1)) / 32.0f, fy = (float)(Y & (INTER_TAB - 1)) / 32.0f;
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
        const float iw = bilin_f(img, h, w, sx, sy, w00, w01, w10, w11);
        const float gxw = bilin_f(gx, h, w, sx, sy, w00, w01, w10, w11), gyw = bilin_f(gy, h, w, sx, sy, w00, w01, w10, w11);
        int Xn, Yn;
        src_coord(a, x, y, AB_SCALE / 2, AB_BITS, Xn, Yn);                    // INTER_NEAREST
        const bool in = Xn >= 0 && Xn < w && Yn >= 0 && Yn < h && premask[(size_t)Yn * w + Xn] != 0;
        const float hatx = -((float)x * h1) - ((float)y * h0), haty = ((float)x * h0) - ((float)y * h1);
        const double j0 = (double)(gxw * hatx + gyw * haty), j1 = (double)gxw, j2 = (double)gyw;
        const double i_ = (double)iw, t_ = (double)tpl[p];
        acc[6] += j0 * j0; acc[7] += j0 * j1; acc[8] += j0 * j2; acc[9] += j1 * j1; acc[10] += j1 * j2; acc[11] += j2 * j2;
        acc[12] += j0 * i_; acc[13] += j1 * i_; acc[14] += j2 * i_;
        if (in) {
            acc[0] += 1.0; acc[1] += i_; acc[2] += i_ * i_; acc[3] += t_; acc[4] += t_ * t_; acc[5] += t_ * i_;
            acc[15] += j0; acc[16] += j1; acc[17] += j2;
            acc[18] += j0 * t_; acc[19] += j1 * t_; acc[20] += j2 * t_;
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < ECC_NSUM; i++) {
        double v = wave_sum(acc[i]);
        if (lane == 0) red[wid][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < ECC_NSUM) {
        double v = 0.0;
        for (int k = 0; k < ECC_T / 64; k++) v += red[k][threadIdx.x];
        partial[(b * gridDim.x + blockIdx.x) * ECC_NSUM + threadIdx.x] = v;
    }
}

// part 2 (one wave per frame): combine the partials in a fixed order, rho, lambda, deltaP = H^-1 * projection, warp update
__global__ void k_ecc_update(EccState *__restrict__ st_all, const double *__restrict__ partial, int nblocks, int max_iters, double eps)
{
    __shared__ double S[ECC_NSUM];
    const size_t b = blockIdx.x;
    EccState &st = st_all[b];
    if (st.done) return;
    if (threadIdx.x < ECC_NSUM) {
        double v = 0.0;
        for (int k = 0; k < nblocks; k++) v += partial[(b * nblocks + k) * ECC_NSUM + threadIdx.x];
        S[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    st.iter++;
    const double n = S[0], mi = S[1] / n, mt = S[3] / n;
    const double img_norm = sqrt(fmax(S[2] - n * mi * mi, 0.0)), tmp_norm = sqrt(fmax(S[4] - n * mt * mt, 0.0));
    const double corr = S[5] - n * mt * mi;
    const double rho = corr / (img_norm * tmp_norm);
    st.last_rho = st.rho;
    st.rho = rho;
    bool fail = !(rho == rho) || !(n > 0.0);
    double H[3][3] = {{S[6], S[7], S[8]}, {S[7], S[9], S[10]}, {S[8], S[10], S[11]}};
    double ip[3], tp[3];
    for (int a = 0; a < 3; a++) { ip[a] = S[12 + a] - mi * S[15 + a]; tp[a] = S[18 + a] - mt * S[15 + a]; }
    // inverse of the symmetric 3x3 Hessian (adjugate)
    const double c00 = H[1][1] * H[2][2] - H[1][2] * H[2][1], c01 = H[0][2] * H[2][1] - H[0][1] * H[2][2], c02 = H[0][1] * H[1][2] - H[0][2] * H[1][1];
    const double det = H[0][0] * c00 + H[1][0] * c01 + H[2][0] * c02;
    if (!(fabs(det) > 0.0)) fail = true;
    const double Hi[3][3] = {{c00 / det, c01 / det, c02 / det},
                             {(H[1][2] * H[2][0] - H[1][0] * H[2][2]) / det, (H[0][0] * H[2][2] - H[0][2] * H[2][0]) / det, (H[0][2] * H[1][0] - H[0][0] * H[1][2]) / det},
                             {(H[1][0] * H[2][1] - H[1][1] * H[2][0]) / det, (H[0][1] * H[2][0] - H[0][0] * H[2][1]) / det, (H[0][0] * H[1][1] - H[0][1] * H[1][0]) / det}};
    double iph[3];
    for (int a = 0; a < 3; a++) iph[a] = Hi[a][0] * ip[0] + Hi[a][1] * ip[1] + Hi[a][2] * ip[2];
    const double lam_n = img_norm * img_norm - (ip[0] * iph[0] + ip[1] * iph[1] + ip[2] * iph[2]);
    const double lam_d = corr - (tp[0] * iph[0] + tp[1] * iph[1] + tp[2] * iph[2]);
    if (!(lam_d > 0.0)) fail = true;
    if (fail) { st.failed = 1; st.done = 1; return; }
    const double lam = lam_n / lam_d;
    double ep[3], dp[3];
    for (int a = 0; a < 3; a++) ep[a] = lam * tp[a] - ip[a];
    for (int a = 0; a < 3; a++) dp[a] = Hi[a][0] * ep[0] + Hi[a][1] * ep[1] + Hi[a][2] * ep[2];
    const float theta = asinf(st.warp[3]) + (float)dp[0];
    st.warp[2] += (float)dp[1];
    st.warp[5] += (float)dp[2];
    st.warp[0] = st.warp[4] = cosf(theta);
    st.warp[3] = sinf(theta);
    st.warp[1] = -st.warp[3];
    if (st.iter >= max_iters || fabs(st.rho - st.last_rho) < eps) st.done = 1;
}

__global__ void k_ecc_init(EccState *st, int B, double eps)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    EccState s;
    s.warp[0] = 1.f; s.warp[1] = 0.f; s.warp[2] = 0.f; s.warp[3] = 0.f; s.warp[4] = 1.f; s.warp[5] = 0.f;
    s.rho = -1.0; s.last_rho = -eps; s.iter = 0; s.done = 0; s.failed = 0; s.pad = 0;
    st[b] = s;
}

// final aligned crop: warpAffine(mov u8, warp, INTER_LINEAR | WARP_INVERSE_MAP, BORDER_REFLECT), or the input when ECC failed
__global__ void k_ecc_apply(const uint8_t *__restrict__ mov_all, uint8_t *__restrict__ out_all, const EccState *__restrict__ st_all, double *__restrict__ info,
                            int h, int w)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    size_t b = blockIdx.z;
    const EccState &st = st_all[b];
    if (x == 0 && y == 0) {
        double *o = info + b * VISTAF_ALIGN_NINFO;
        const float id[6] = {1, 0, 0, 0, 1, 0};
        for (int i = 0; i < 6; i++) o[VISTAF_AI_WARP + i] = st.failed ? (double)id[i] : (double)st.warp[i];
        o[VISTAF_AI_RHO] = st.failed ? __longlong_as_double(0x7ff8000000000000ll) : st.rho;
        o[VISTAF_AI_ECC_ITERS] = (double)st.iter;
        o[VISTAF_AI_ECC_FAILED] = (double)st.failed;
    }
    if (x >= w) return;
    const uint8_t *mov = mov_all + b * (size_t)h * w;
    uint8_t *out = out_all + b * (size_t)h * w;
    if (st.failed) { out[(size_t)y * w + x] = mov[(size_t)y * w + x]; return; }
    Aff a;
    for (int i = 0; i < 6; i++) a.m[i] = (double)st.warp[i];
    int X, Y;
    src_coord(a, x, y, AB_SCALE / INTER_TAB / 2, AB_BITS - INTER_BITS, X, Y);
    const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS, ax = X & (INTER_TAB - 1), ay = Y & (INTER_TAB - 1);
    const int xa = reflect_b(sx, w), xb = reflect_b(sx + 1, w), ya = reflect_b(sy, h), yb = reflect_b(sy + 1, h);
    const int w00 = (INTER_TAB - ax) * (INTER_TAB - ay) * 32, w01 = ax * (INTER_TAB - ay) * 32, w10 = (INTER_TAB - ax) * ay * 32, w11 = ax * ay * 32;
    int acc = mov[(size_t)ya * w + xa] * w00 + mov[(size_t)ya * w + xb] * w01 + mov[(size_t)yb * w + xa] * w10 + mov[(size_t)yb * w + xb] * w11;
    int v = (acc + (1 << 14)) >> 15;
    out[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ void k_circle_mask(uint8_t *m, int h, int w, int cx, int cy, int r)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    m[(size_t)y * w + x] = (uint8_t)(((x - cx) * (x - cx) + (y - cy) * (y - cy)) <= r * r);
}

int optimal_dft(int n)
{
    for (int m = n;; m++) {
        int k = m;
        for (int p : {2, 3, 5}) while (k % p == 0) k /= p;
        if (k == 1) return m;
    }
}

int gauss_taps(double sigma, std::vector<float> &f)
{
    int n = ((int)lrint(sigma * 4 * 2 + 1)) | 1;         // cv::GaussianBlur ksize rule, CV_32F
    std::vector<double> t(n);
    double s2 = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; i++) { double x = i - (n - 1) * 0.5; t[i] = std::exp(s2 * x * x); sum += t[i]; }
    f.resize(n);
    for (int i = 0; i < n; i++) f[i] = (float)(t[i] * (1.0 / sum));
    return n;
}

}  // namespace

struct vistaf_align_handle {
    vistaf_align_config cfg;
    int H, W, M, N, cx, cy, r, maxB;
    int x1, y1, x2, y2, ch, cw, cxl, cyl, rl;
    std::vector<void *> allocs;
    float *g7 = nullptr, *g5 = nullptr;
    int k7 = 0, k5 = 0;
    float *gray_f = nullptr, *tmp_f = nullptr, *pad_f = nullptr, *corr = nullptr;        // full-frame planes
    float2 *F_ref = nullptr, *F_def = nullptr;
    uint8_t *crop_bgr = nullptr;
    unsigned long long *peak_part = nullptr;
    hipfftHandle plan_r2c = 0, plan_c2r = 0;
    bool have_plans = false, have_ref = false;
    uint8_t *ref_gray = nullptr, *circ = nullptr, *mov_u8 = nullptr;                         // crops
    float *tpl = nullptr, *mov_f = nullptr, *mov_tmp = nullptr, *gx = nullptr, *gy = nullptr, *crop_tmp = nullptr;
    EccState *st = nullptr;
    double *partial = nullptr;
    int *done_host = nullptr;
};

template <typename T>
static int aalloc(vistaf_align_handle *h, T **p, size_t n)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
    if (e != hipSuccess) return set_error(VISTAF_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

extern "C" {

void vistaf_align_default_config(vistaf_align_config *c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->apply_global_shift = 1; c->use_ecc = 1; c->ecc_iters = 300; c->ecc_eps = 1e-7; c->ecc_gauss_sigma = 5.0; c->shift_blur_sigma = 7.0;
}

void vistaf_align_destroy(vistaf_align_handle *h)
{
    if (!h) return;
    if (h->have_plans) { hipfftDestroy(h->plan_r2c); hipfftDestroy(h->plan_c2r); }
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
}

int vistaf_align_create(const vistaf_align_config *cfg, int H, int W, int cx, int cy, int r, int max_batch, vistaf_align_handle **out)
{
    if (!cfg || !out) return set_error(VISTAF_E_INVALID, "null argument");
    if (H < 16 || W < 16 || r < 4 || max_batch < 1) return set_error(VISTAF_E_INVALID, "bad geometry");
    if (!(cfg->shift_blur_sigma > 0) || cfg->ecc_iters < 1) return set_error(VISTAF_E_INVALID, "bad alignment config");
    vistaf_align_handle *h = new vistaf_align_handle();
    h->cfg = *cfg; h->H = H; h->W = W; h->cx = cx; h->cy = cy; h->r = r; h->maxB = max_batch;
    h->M = optimal_dft(H); h->N = optimal_dft(W);
    // ROI bounding box and local circle (shape_ftp.py:1502-1519)
    h->x1 = std::max(0, cx - r); h->x2 = std::min(W, cx + r); h->y1 = std::max(0, cy - r); h->y2 = std::min(H, cy + r);
    h->cw = h->x2 - h->x1; h->ch = h->y2 - h->y1;
    if (h->cw < 8 || h->ch < 8) { delete h; return set_error(VISTAF_E_INVALID, "ROI outside the frame"); }
    h->cxl = cx - h->x1; h->cyl = cy - h->y1;
    h->rl = std::min(std::min(r, h->cxl), std::min(std::min(h->cyl, h->cw - 1 - h->cxl), h->ch - 1 - h->cyl));
#define TRYA(x) do { int rc_ = (x); if (rc_) { vistaf_align_destroy(h); return rc_; } } while (0)
    std::vector<float> t7, t5;
    h->k7 = gauss_taps(cfg->shift_blur_sigma, t7);
    TRYA(aalloc(h, &h->g7, t7.size()));
    if (hipMemcpy(h->g7, t7.data(), t7.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { vistaf_align_destroy(h); return set_error(VISTAF_E_HIP, "memcpy taps"); }
    if (cfg->ecc_gauss_sigma > 0) {
        h->k5 = gauss_taps(cfg->ecc_gauss_sigma, t5);
        TRYA(aalloc(h, &h->g5, t5.size()));
        if (hipMemcpy(h->g5, t5.data(), t5.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { vistaf_align_destroy(h); return set_error(VISTAF_E_HIP, "memcpy taps"); }
    }
    if (h->k7 > 511 || h->k5 > 511) { vistaf_align_destroy(h); return set_error(VISTAF_E_INVALID, "gaussian sigma too large"); }
    const size_t FP = (size_t)H * W, MP = (size_t)h->M * h->N, HP = (size_t)h->M * (h->N / 2 + 1), CP = (size_t)h->ch * h->cw;
    TRYA(aalloc(h, &h->gray_f, FP)); TRYA(aalloc(h, &h->tmp_f, FP)); TRYA(aalloc(h, &h->pad_f, MP)); TRYA(aalloc(h, &h->corr, MP));
    TRYA(aalloc(h, &h->F_ref, HP)); TRYA(aalloc(h, &h->F_def, HP));
    TRYA(aalloc(h, &h->crop_bgr, CP * 3));
    TRYA(aalloc(h, &h->peak_part, 256));
    TRYA(aalloc(h, &h->ref_gray, CP)); TRYA(aalloc(h, &h->circ, CP)); TRYA(aalloc(h, &h->tpl, CP)); TRYA(aalloc(h, &h->crop_tmp, CP * max_batch));
    TRYA(aalloc(h, &h->mov_u8, CP * max_batch)); TRYA(aalloc(h, &h->mov_f, CP * max_batch)); TRYA(aalloc(h, &h->mov_tmp, CP * max_batch));
    TRYA(aalloc(h, &h->gx, CP * max_batch)); TRYA(aalloc(h, &h->gy, CP * max_batch));
    TRYA(aalloc(h, &h->st, (size_t)max_batch)); TRYA(aalloc(h, &h->partial, (size_t)max_batch * ECC_BLOCKS * ECC_NSUM));
    if (hipfftPlan2d(&h->plan_r2c, h->M, h->N, HIPFFT_R2C) != HIPFFT_SUCCESS || hipfftPlan2d(&h->plan_c2r, h->M, h->N, HIPFFT_C2R) != HIPFFT_SUCCESS) {
        vistaf_align_destroy(h);
        return set_error(VISTAF_E_HIP, "hipfftPlan2d failed");
    }
    h->have_plans = true;
    *out = h;
    return 0;
}

int vistaf_align_geometry(const vistaf_align_handle *h, int32_t *x1, int32_t *y1, int32_t *x2, int32_t *y2, int32_t *crop_h, int32_t *crop_w,
                          int32_t *cx_local, int32_t *cy_local, int32_t *r_local)
{
    if (!h) return set_error(VISTAF_E_INVALID, "null handle");
    if (x1) *x1 = h->x1; if (y1) *y1 = h->y1; if (x2) *x2 = h->x2; if (y2) *y2 = h->y2;
    if (crop_h) *crop_h = h->ch; if (crop_w) *crop_w = h->cw;
    if (cx_local) *cx_local = h->cxl; if (cy_local) *cy_local = h->cyl; if (r_local) *r_local = h->rl;
    return 0;
}

}  // extern "C"

// windowed spectrum of the blurred grey frame (GaussianBlur sigma 7, Hanning window, zero padding to the optimal DFT size)
static int frame_spectrum(vistaf_align_handle *h, const uint8_t *d_bgr, float2 *F, hipStream_t st)
{
    const size_t FP = (size_t)h->H * h->W;
    hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)((FP + 255) / 256)), dim3(256), 0, st, d_bgr, (uint8_t *)nullptr, h->gray_f, FP, h->cfg.gray_coeffs);
    launch_gauss_rows(h->gray_f, h->tmp_f, h->g7, h->k7, 1, h->H, h->W, st);
    launch_gauss_cols(h->tmp_f, h->gray_f, h->g7, h->k7, 1, h->H, h->W, st);
    hipLaunchKernelGGL(k_hann_mul, dim3((h->W + 255) / 256, h->H), dim3(256), 0, st, h->gray_f, h->H, h->W);
    const float *src = h->gray_f;
    if (h->M != h->H || h->N != h->W) {
        ACHK(hipMemsetAsync(h->pad_f, 0, (size_t)h->M * h->N * sizeof(float), st));
        ACHK(hipMemcpy2DAsync(h->pad_f, (size_t)h->N * sizeof(float), h->gray_f, (size_t)h->W * sizeof(float), (size_t)h->W * sizeof(float), h->H, hipMemcpyDeviceToDevice, st));
        src = h->pad_f;
    }
    FCHK(hipfftSetStream(h->plan_r2c, st));
    FCHK(hipfftExecR2C(h->plan_r2c, (hipfftReal *)src, (hipfftComplex *)F));
    return 0;
}

extern "C" {

int vistaf_align_set_reference(vistaf_align_handle *h, const uint8_t *d_ref_bgr, uint8_t *d_ref_gray_crop, void *stream)
{
    if (!h || !d_ref_bgr) return set_error(VISTAF_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    int rc = frame_spectrum(h, d_ref_bgr, h->F_ref, st);
    if (rc) return rc;
    const size_t CP = (size_t)h->ch * h->cw;
    hipLaunchKernelGGL(k_crop_bgr, dim3((h->cw + 255) / 256, h->ch), dim3(256), 0, st, d_ref_bgr, h->crop_bgr, h->W, h->x1, h->y1, h->cw);
    hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)((CP + 255) / 256)), dim3(256), 0, st, h->crop_bgr, h->ref_gray, (float *)nullptr, CP, h->cfg.gray_coeffs);
    hipLaunchKernelGGL(k_circle_mask, dim3((h->cw + 255) / 256, h->ch), dim3(256), 0, st, h->circ, h->ch, h->cw, h->cxl, h->cyl, h->rl);
    // ECC template: ref / 255, GaussianBlur(ecc_gauss)
    hipLaunchKernelGGL(k_u8_to_unit, dim3((unsigned)((CP + 255) / 256)), dim3(256), 0, st, h->ref_gray, h->tpl, CP);
    if (h->k5) {
        launch_gauss_rows(h->tpl, h->crop_tmp, h->g5, h->k5, 1, h->ch, h->cw, st);
        launch_gauss_cols(h->crop_tmp, h->tpl, h->g5, h->k5, 1, h->ch, h->cw, st);
    }
    if (d_ref_gray_crop) ACHK(hipMemcpyAsync(d_ref_gray_crop, h->ref_gray, CP, hipMemcpyDeviceToDevice, st));
    ACHK(hipStreamSynchronize(st));
    h->have_ref = true;
    return 0;
}

int vistaf_align_batch(vistaf_align_handle *h, const uint8_t *d_def_bgr, int B, uint8_t *d_out, double *d_info, void *stream)
{
    if (!h || !d_def_bgr || !d_out || !d_info) return set_error(VISTAF_E_INVALID, "null argument");
    if (!h->have_ref) return set_error(VISTAF_E_STATE, "vistaf_align_set_reference has not been called");
    if (B < 1 || B > h->maxB) return set_error(VISTAF_E_STATE, "batch exceeds max_batch");
    hipStream_t st = (hipStream_t)stream;
    const size_t FP = (size_t)h->H * h->W, CP = (size_t)h->ch * h->cw, HP = (size_t)h->M * (h->N / 2 + 1);
    ACHK(hipMemsetAsync(d_info, 0, sizeof(double) * VISTAF_ALIGN_NINFO * B, st));
    std::vector<double> shifts(3 * (size_t)B, 0.0);
    for (int b = 0; b < B; b++) {
        const uint8_t *bgr = d_def_bgr + (size_t)b * FP * 3;
        double *info = d_info + (size_t)b * VISTAF_ALIGN_NINFO;
        // ---- estimate_global_shift (:529-535)
        int rc = frame_spectrum(h, bgr, h->F_def, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_cross_power, dim3((unsigned)((HP + 255) / 256)), dim3(256), 0, st, h->F_ref, h->F_def, h->F_def, HP);
        FCHK(hipfftSetStream(h->plan_c2r, st));
        FCHK(hipfftExecC2R(h->plan_c2r, (hipfftComplex *)h->F_def, (hipfftReal *)h->corr));
        hipLaunchKernelGGL(k_peak_partial, dim3(256), dim3(1024), 0, st, h->corr, h->M, h->N, h->peak_part);
        hipLaunchKernelGGL(k_peak_final, dim3(1), dim3(64), 0, st, h->corr, h->M, h->N, h->peak_part, 256, info);
        // ---- warpAffine of the ROI window only (the rest of the shifted frame is never looked at), then BGR2GRAY
        Aff a = {{1, 0, 0, 0, 1, 0}};
        if (h->cfg.apply_global_shift) {
            ACHK(hipMemcpyAsync(&shifts[3 * (size_t)b], info, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            ACHK(hipStreamSynchronize(st));
            // M = [[1,0,dx],[0,1,dy]] as float32, inverted in double (cv::invertAffineTransform)
            const double dx = (double)(float)shifts[3 * (size_t)b], dy = (double)(float)shifts[3 * (size_t)b + 1];
            a.m[2] = -dx; a.m[5] = -dy;
            hipLaunchKernelGGL(k_warp_u8<3>, dim3((h->cw + 255) / 256, h->ch), dim3(256), 0, st, bgr, h->crop_bgr, a, h->H, h->W, h->x1, h->y1, h->x2, h->y2);
        } else {
            hipLaunchKernelGGL(k_crop_bgr, dim3((h->cw + 255) / 256, h->ch), dim3(256), 0, st, bgr, h->crop_bgr, h->W, h->x1, h->y1, h->cw);
        }
        hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)((CP + 255) / 256)), dim3(256), 0, st, h->crop_bgr, h->mov_u8 + (size_t)b * CP, (float *)nullptr, CP, h->cfg.gray_coeffs);
    }
    // ---- align_crop_ecc (:549-578), all frames together
    hipLaunchKernelGGL(k_ecc_init, dim3((B + 63) / 64), dim3(64), 0, st, h->st, B, h->cfg.ecc_eps);
    if (h->cfg.use_ecc) {
        hipLaunchKernelGGL(k_u8_to_unit, dim3((unsigned)((CP * B + 255) / 256)), dim3(256), 0, st, h->mov_u8, h->mov_f, CP * B);
        if (h->k5) {
            launch_gauss_rows(h->mov_f, h->mov_tmp, h->g5, h->k5, B, h->ch, h->cw, st);
            launch_gauss_cols(h->mov_tmp, h->mov_f, h->g5, h->k5, B, h->ch, h->cw, st);
        }
        hipLaunchKernelGGL(k_ecc_grad, dim3((h->cw + 255) / 256, h->ch, B), dim3(256), 0, st, h->mov_f, h->circ, h->gx, h->gy, h->ch, h->cw);
        std::vector<EccState> hs((size_t)B);
        for (int it = 0; it < h->cfg.ecc_iters; it++) {
            hipLaunchKernelGGL(k_ecc_accumulate, dim3(ECC_BLOCKS, B), dim3(ECC_T), 0, st, h->tpl, h->mov_f, h->gx, h->gy, h->circ, h->st, h->partial, h->ch, h->cw);
            hipLaunchKernelGGL(k_ecc_update, dim3(B), dim3(64), 0, st, h->st, h->partial, ECC_BLOCKS, h->cfg.ecc_iters, h->cfg.ecc_eps);
            if ((it & 7) == 7) {                                   // all frames converged?  (the iteration count is data dependent)
                ACHK(hipMemcpyAsync(hs.data(), h->st, sizeof(EccState) * B, hipMemcpyDeviceToHost, st));
                ACHK(hipStreamSynchronize(st));
                bool all = true;
                for (int b = 0; b < B; b++) all = all && hs[b].done;
                if (all) break;
            }
        }
        hipLaunchKernelGGL(k_ecc_apply, dim3((h->cw + 255) / 256, h->ch, B), dim3(256), 0, st, h->mov_u8, d_out, h->st, d_info, h->ch, h->cw);
    } else {
        ACHK(hipMemcpyAsync(d_out, h->mov_u8, CP * B, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_ecc_apply, dim3(1, 1, B), dim3(1), 0, st, h->mov_u8, h->mov_u8, h->st, d_info, 0, 0);   // identity warps into the records
    }
    ACHK(hipStreamSynchronize(st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(VISTAF_E_HIP, std::string("launch: ") + hipGetErrorString(e));
    return 0;
}

}  // extern "C"
