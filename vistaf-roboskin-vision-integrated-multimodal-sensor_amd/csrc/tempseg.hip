// Periodic-stripe segmentation of the temperature modality (include/vistaf_temp.h): Code/temperature_sensor.py:437-540 on the GPU.
//
//   gray = BGR2GRAY; sat = dilate(gray >= 245 & roi, ellipse 13) & roi; roi_eff = roi & ~sat                       (:378-387, :441-446)
//   g = gray, median(gray[roi_eff]) outside roi; I = (g / blur_20(g)) / mean(.[roi_eff])                             (:448-452, :363-375)
//   carrier = strongest of the top-16 spectrum peaks in the right half-plane near the centre row                      (:454-461)
//   z = ifft2(fft2(I) restricted to the disc of radius 22 around the carrier); phi0 = angle(sum_roi z (I - 1))        (:463-476)
//   A = Re(z e^{-i phi0}) >= 0; the darker of A / not-A is "dark"; close 3x31, open 3x7                               (:478-499, :390-406)
// The spectrum for the carrier SEARCH is hipFFT's (float32; only peak positions are read off it).  The band-pass itself never forms the
// full spectrum: the (2R+1)^2 bins around the carrier are computed and transformed back by the path's pruned float64 DFT (k_dft*.hip, stages
// 1 and 4 on the matrix cores), so z carries float64 rounding only, as upstream's complex128 field does.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vistaf_ftp.h"
#include "../../include/vistaf_temp.h"
#include "kernels.hpp"

namespace vf { int set_error(int code, const std::string &msg); }
using namespace vf;

#define TCHK(x)                                                                                                  \
    do {                                                                                                         \
        hipError_t e_ = (x);                                                                                     \
        if (e_ != hipSuccess) return set_error(VISTAF_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));   \
    } while (0)

namespace {

__global__ void k_ts_sat0(const float *__restrict__ gray, const uint8_t *__restrict__ roi, float thr, uint8_t *__restrict__ sat, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sat[i] = (uint8_t)(roi[i] && gray[i] >= thr);
}
__global__ void k_ts_and_not(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)(a[i] && !b[i]);
}
__global__ void k_ts_and(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)(a[i] && b[i]);
}
// g[~roi_full] = med (:450)
__global__ void k_ts_fill(const float *__restrict__ gray, const uint8_t *__restrict__ roi, const float *__restrict__ med, float *__restrict__ g, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = roi[i] ? gray[i] : med[0];
}
// blur[blur < 1e-6] = 1; norm = g / blur (:370-372); without the Gaussian norm = g
__global__ void k_ts_norm(const float *__restrict__ g, const float *__restrict__ blur, float *__restrict__ norm, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float bl = blur ? blur[i] : 1.0f;
    if (blur && bl < 1e-6f) bl = 1.0f;
    norm[i] = blur ? __fdiv_rn(g[i], bl) : g[i];
}
// deterministic two-level sums over a mask: partial[blockIdx] then one block adds the partials in index order
__global__ __launch_bounds__(256) void k_ts_masked_sum(const float *__restrict__ v, const uint8_t *__restrict__ mask, double *__restrict__ partial, size_t n)
{
    __shared__ double sd[16];
    double acc = 0.0, cnt = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (mask[i]) { acc += (double)v[i]; cnt += 1.0; }
    acc = block_sum<double>(acc, sd);
    cnt = block_sum<double>(cnt, sd);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = acc; partial[2 * blockIdx.x + 1] = cnt; }
}
__global__ void k_ts_final2(const double *__restrict__ partial, int nblocks, int stride, double *__restrict__ out)
{
    if ((int)threadIdx.x >= stride) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; b++) s += partial[(size_t)b * stride + threadIdx.x];
    out[threadIdx.x] = s;
}
// I = norm / mu, mu = mean(norm[roi_eff]) rounded to float32 (|mu| <= 1e-9 -> 1) (:373-375)
__global__ void k_ts_scale(const float *__restrict__ norm, const double *__restrict__ sums, float *__restrict__ inorm, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float mu = sums[1] > 0.0 ? (float)(sums[0] / sums[1]) : 1.0f;
    if (!(fabsf(mu) > 1e-9f)) mu = 1.0f;
    inorm[i] = __fdiv_rn(norm[i], mu);
}
// |fftshift(fft2(I))| from the Hermitian half hipFFT returns
__global__ void k_ts_mag_full(const float2 *__restrict__ F, double *__restrict__ mag, int H, int W)
{
    int sx = blockIdx.x * blockDim.x + threadIdx.x, sy = blockIdx.y;
    if (sx >= W) return;
    const int cy = H / 2, cx = W / 2, Wh = W / 2 + 1;
    int fy = (sy - cy + H) % H, fx = (sx - cx + W) % W;
    if (fx > W / 2) { fx = W - fx; fy = (H - fy) % H; }
    const float2 v = F[(size_t)fy * Wh + fx];
    mag[(size_t)sy * W + sx] = sqrt((double)v.x * v.x + (double)v.y * v.y);
}
// c = sum over roi_eff of z * (I - 1) (:471-472), complex128; partial[2*block + {0,1}]
__global__ __launch_bounds__(256) void k_ts_csum(const double2 *__restrict__ z, const float *__restrict__ inorm, const uint8_t *__restrict__ roi_eff,
                                                 double *__restrict__ partial, size_t n)
{
    __shared__ double sd[16];
    double ar = 0.0, ai = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (roi_eff[i]) { const double m = (double)__fsub_rn(inorm[i], 1.0f); const double2 v = z[i]; ar += v.x * m; ai += v.y * m; }
    ar = block_sum<double>(ar, sd);
    ai = block_sum<double>(ai, sd);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = ar; partial[2 * blockIdx.x + 1] = ai; }
}
// s = float32(Re(z e^{-i phi0})); A = s >= 0 & roi_eff; sums of gray and counts over A and over roi_eff & ~A (:474-481)
__global__ __launch_bounds__(256) void k_ts_sign(const double2 *__restrict__ z, double c0, double s0, const uint8_t *__restrict__ roi_eff,
                                                 const float *__restrict__ gray, uint8_t *__restrict__ mask_a, double *__restrict__ partial, size_t n)
{
    __shared__ double sd[16];
    double ga = 0.0, na = 0.0, gb = 0.0, nb = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint8_t a = 0;
        if (roi_eff[i]) {
            const double2 v = z[i];
            // z * exp(-i phi0): real part zr*cos(phi0) - zi*(-sin(phi0)), as NumPy multiplies complex128
            const float s = (float)(v.x * c0 - v.y * (-s0));
            a = s >= 0.0f;
            if (a) { ga += (double)gray[i]; na += 1.0; } else { gb += (double)gray[i]; nb += 1.0; }
        }
        mask_a[i] = a;
    }
    ga = block_sum<double>(ga, sd); na = block_sum<double>(na, sd); gb = block_sum<double>(gb, sd); nb = block_sum<double>(nb, sd);
    if (threadIdx.x == 0) { double *o = partial + 4 * (size_t)blockIdx.x; o[0] = ga; o[1] = na; o[2] = gb; o[3] = nb; }
}

RowSpanSE rect_se(int kx, int ky)
{
    RowSpanSE se;
    se.k = ky;
    for (int i = 0; i < 33; i++) { se.lo[i] = (int8_t)(-(kx / 2)); se.hi[i] = (int8_t)(kx / 2); }
    return se;
}
RowSpanSE ellipse_se(int k)
{
    RowSpanSE se;
    se.k = k;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < 33; i++) { se.lo[i] = 1; se.hi[i] = -1; }
    for (int i = 0; i < k; i++) {
        const int dy = i - r;
        const int dx = (int)std::nearbyint(c * std::sqrt((r * r - dy * dy) * inv_r2));
        const int j1 = std::max(c - dx, 0), j2 = std::min(c + dx + 1, k);
        se.lo[i] = (int8_t)(j1 - c);
        se.hi[i] = (int8_t)(j2 - 1 - c);
    }
    return se;
}
int ensure_odd(int k) { return (k % 2) ? k : k + 1; }

constexpr int TS_RB = 1024;      // blocks of the two-level reductions

}  // namespace

struct vistaf_tempseg_handle {
    vistaf_tempseg_config cfg;
    int H = 0, W = 0, pm = 0;
    size_t P = 0;
    std::vector<void *> allocs;
    float *gray = nullptr, *g = nullptr, *tmpf = nullptr, *blur = nullptr, *norm = nullptr, *inorm = nullptr, *amp = nullptr, *gk = nullptr, *win = nullptr, *med = nullptr;
    int gksize = 0;
    uint8_t *sat0 = nullptr, *sat = nullptr, *roi_eff = nullptr, *ma = nullptr, *mb = nullptr, *m1 = nullptr, *m2 = nullptr;
    uint16_t *prefix = nullptr;
    float2 *F = nullptr;
    double *mag = nullptr, *peaks = nullptr, *partial = nullptr, *sums = nullptr;
    double2 *Ex = nullptr, *Ey = nullptr, *Gx = nullptr, *Gy = nullptr, *T = nullptr, *patch = nullptr, *z = nullptr;
    CarrierGeom *geom = nullptr;
    float *req_med = nullptr;
    int *cnt = nullptr;
    hipfftHandle plan = 0;
    bool have_plan = false;
    uint16_t *gamma_tab = nullptr, *cbrt_tab = nullptr;       // cv::RGB2Lab_b tables
    // map-domain stages (allocated on first use): two float planes, two masks, the whole-frame march's scratch, counters, Gaussian taps
    float *tmA = nullptr, *tmB = nullptr, *tm_kx = nullptr, *tm_ky = nullptr;
    uint8_t *tmM1 = nullptr, *tmM2 = nullptr;
    void *tm_scratch = nullptr;
    uint32_t *tm_stats = nullptr;
    int32_t *tm_status = nullptr;
    unsigned long long *tm_counts = nullptr;
    LabCoef lab;
};

namespace {
template <typename T>
int talloc(vistaf_tempseg_handle *h, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 256);
    if (e != hipSuccess) return set_error(VISTAF_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}
inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }
}  // namespace

extern "C" {

int vistaf_tempseg_default_config(vistaf_tempseg_config *c)
{
    if (!c) return set_error(VISTAF_E_INVALID, "null config");
    memset(c, 0, sizeof(*c));
    c->seg_band_radius = 22; c->seg_dc_exclusion = 28; c->seg_illum_sigma = 20; c->sat_thresh_gray = 245; c->sat_dilate_ksize = 13;
    c->post_close_kx = 3; c->post_close_ky = 31; c->post_open_kx = 3; c->post_open_ky = 7; c->n_peaks = 16; c->seg_peak_max_dy_from_center = 0.14;
    return 0;
}

void vistaf_tempseg_destroy(vistaf_tempseg_handle *h)
{
    if (!h) return;
    for (void *p : h->allocs) hipFree(p);
    if (h->have_plan) hipfftDestroy(h->plan);
    delete h;
}

int vistaf_tempseg_create(const vistaf_tempseg_config *cfg, int H, int W, vistaf_tempseg_handle **out)
{
    if (!cfg || !out) return set_error(VISTAF_E_INVALID, "null argument");
    if (H < 64 || W < 64 || H % 16) return set_error(VISTAF_E_INVALID, "frame height must be a multiple of 16 (matrix-core strips), both sides >= 64");
    const int R = cfg->seg_band_radius;
    if (R < 1 || 2 * R + 1 > 127 || cfg->n_peaks < 1 || cfg->n_peaks > 64) return set_error(VISTAF_E_INVALID, "band radius / peak count out of range");
    for (int k : {ensure_odd(std::max(1, cfg->post_close_ky)), ensure_odd(std::max(1, cfg->post_open_ky)), ensure_odd(cfg->sat_dilate_ksize)})
        if (k > 33) return set_error(VISTAF_E_INVALID, "structuring element taller than 33");
    for (int k : {ensure_odd(std::max(1, cfg->post_close_kx)), ensure_odd(std::max(1, cfg->post_open_kx))})
        if (k > 127) return set_error(VISTAF_E_INVALID, "structuring element wider than 127");
    vistaf_tempseg_handle *h = new vistaf_tempseg_handle();
    h->cfg = *cfg; h->H = H; h->W = W; h->P = (size_t)H * W; h->pm = 2 * R + 1;
    const size_t P = h->P;
    const int pm = h->pm;
    int rc = 0;
#define TRY(x) do { rc = (x); if (rc) { vistaf_tempseg_destroy(h); return rc; } } while (0)
    TRY(talloc(h, &h->gray, P)); TRY(talloc(h, &h->g, P)); TRY(talloc(h, &h->tmpf, P)); TRY(talloc(h, &h->blur, P)); TRY(talloc(h, &h->norm, P));
    TRY(talloc(h, &h->inorm, P)); TRY(talloc(h, &h->amp, P)); TRY(talloc(h, &h->med, 4)); TRY(talloc(h, &h->cnt, 4));
    TRY(talloc(h, &h->sat0, P)); TRY(talloc(h, &h->sat, P)); TRY(talloc(h, &h->roi_eff, P)); TRY(talloc(h, &h->ma, P)); TRY(talloc(h, &h->mb, P));
    TRY(talloc(h, &h->m1, P)); TRY(talloc(h, &h->m2, P)); TRY(talloc(h, &h->prefix, P));
    TRY(talloc(h, &h->F, (size_t)H * (W / 2 + 1))); TRY(talloc(h, &h->mag, P)); TRY(talloc(h, &h->peaks, 192));
    TRY(talloc(h, &h->partial, (size_t)4 * TS_RB)); TRY(talloc(h, &h->sums, 8));
    TRY(talloc(h, &h->Ex, (size_t)W * pm)); TRY(talloc(h, &h->Gx, (size_t)W * pm)); TRY(talloc(h, &h->Ey, (size_t)H * pm)); TRY(talloc(h, &h->Gy, (size_t)H * pm));
    TRY(talloc(h, &h->T, (size_t)std::max(H, W) * pm)); TRY(talloc(h, &h->patch, (size_t)pm * pm)); TRY(talloc(h, &h->z, P)); TRY(talloc(h, &h->geom, 1));
    TRY(talloc(h, &h->win, (size_t)pm * pm)); TRY(talloc(h, &h->req_med, 1));
    {
        const float neg = -1.0f;                                    // launch_select: a negative request is the median
        if (hipMemcpy(h->req_med, &neg, sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { vistaf_tempseg_destroy(h); return set_error(VISTAF_E_HIP, "memcpy"); }
        std::vector<float> win((size_t)pm * pm);                    // the band-pass disc (:463-465) as the patch "window"
        for (int a = 0; a < pm; a++)
            for (int c = 0; c < pm; c++) win[(size_t)a * pm + c] = ((a - R) * (a - R) + (c - R) * (c - R) <= R * R) ? 1.0f : 0.0f;
        if (hipMemcpy(h->win, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { vistaf_tempseg_destroy(h); return set_error(VISTAF_E_HIP, "memcpy"); }
    }
    if (cfg->seg_illum_sigma > 0) {
        const double sigma = (double)cfg->seg_illum_sigma;
        const int n = ((int)std::nearbyint(sigma * 4 * 2 + 1)) | 1;    // cv::GaussianBlur ksize rule, CV_32F
        if (n > 511) { vistaf_tempseg_destroy(h); return set_error(VISTAF_E_INVALID, "illumination sigma too large"); }
        std::vector<double> t(n);
        double s2 = -0.5 / (sigma * sigma), sum = 0;
        for (int i = 0; i < n; i++) { double x = i - (n - 1) * 0.5; t[i] = std::exp(s2 * x * x); sum += t[i]; }
        std::vector<float> f(n);
        for (int i = 0; i < n; i++) f[i] = (float)(t[i] * (1.0 / sum));
        TRY(talloc(h, &h->gk, (size_t)n));
        if (hipMemcpy(h->gk, f.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { vistaf_tempseg_destroy(h); return set_error(VISTAF_E_HIP, "memcpy"); }
        h->gksize = n;
    }
    {
        // The integer tables of OpenCV's 8-bit BGR2LAB: sRGB gamma of i / 255 scaled by 255 * 2^3, the Lab cube-root function of
        // i / (255 * 2^3) scaled by 2^15, and the sRGB -> XYZ (D65) matrix over the white point scaled by 2^12.
        std::vector<uint16_t> gt(256), ct(LAB_CBRT_N);
        for (int i = 0; i < 256; i++) {
            const double x = (double)((float)i / 255.0f);
            const double g = x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
            gt[i] = (uint16_t)std::nearbyint(255.0 * (1 << LAB_GAMMA_SHIFT) * g);
        }
        for (int i = 0; i < LAB_CBRT_N; i++) {
            const double x = (double)i / (255.0 * (1 << LAB_GAMMA_SHIFT));
            const double f = x < 0.008856 ? x * 7.787 + 0.13793103448275862 : std::cbrt(x);
            ct[i] = (uint16_t)std::nearbyint((double)(1 << LAB_SHIFT2) * f);
        }
        const double m[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
        const double wp[3] = {0.950456, 1.0, 1.088754};
        for (int i = 0; i < 9; i++) h->lab.c[i] = (int)std::nearbyint((double)(1 << LAB_SHIFT) * m[i] / wp[i / 3]);
        h->lab.lscale = (116 * 255 + 50) / 100;
        h->lab.lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
        TRY(talloc(h, &h->gamma_tab, (size_t)256)); TRY(talloc(h, &h->cbrt_tab, (size_t)LAB_CBRT_N));
        if (hipMemcpy(h->gamma_tab, gt.data(), gt.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->cbrt_tab, ct.data(), ct.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { vistaf_tempseg_destroy(h); return set_error(VISTAF_E_HIP, "memcpy"); }
    }
    if (hipfftPlan2d(&h->plan, H, W, HIPFFT_R2C) != HIPFFT_SUCCESS) { vistaf_tempseg_destroy(h); return set_error(VISTAF_E_HIP, "hipfftPlan2d failed"); }
    h->have_plan = true;
#undef TRY
    *out = h;
    return 0;
}

int vistaf_tempseg_segment(vistaf_tempseg_handle *h, const uint8_t *d_bgr, const uint8_t *d_roi, uint8_t *d_dark, uint8_t *d_light, uint8_t *d_roi_eff,
                           uint8_t *d_sat, double *info, void *stream)
{
    if (!h || !d_bgr || !d_roi) return set_error(VISTAF_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const vistaf_tempseg_config &c = h->cfg;
    const int H = h->H, W = h->W, pm = h->pm, R = c.seg_band_radius;
    const size_t P = h->P;
    // ---- gray, saturation mask, effective ROI (:378-387, :441-446)
    launch_to_gray(d_bgr, VISTAF_FMT_BGR_U8, h->gray, 1, (int)P, st);
    hipLaunchKernelGGL(k_ts_sat0, grid1(P), dim3(256), 0, st, h->gray, d_roi, (float)c.sat_thresh_gray, h->sat0, P);
    const int ks = ensure_odd(c.sat_dilate_ksize);
    if (ks > 1) {
        launch_morph(h->sat0, h->m1, 1, H, W, ellipse_se(ks), true, nullptr, nullptr, st, h->prefix);
        hipLaunchKernelGGL(k_ts_and, grid1(P), dim3(256), 0, st, h->m1, d_roi, h->sat, P);
    } else TCHK(hipMemcpyAsync(h->sat, h->sat0, P, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_ts_and_not, grid1(P), dim3(256), 0, st, d_roi, h->sat, h->roi_eff, P);
    // ---- median fill, illumination normalisation (:448-452, :363-375)
    launch_select(h->gray, h->roi_eff, P, nullptr, false, h->req_med, 1, h->med, h->cnt, 1, (int)P, st);
    hipLaunchKernelGGL(k_ts_fill, grid1(P), dim3(256), 0, st, h->gray, d_roi, h->med, h->g, P);
    if (h->gksize) launch_gauss_blur(h->g, h->tmpf, h->blur, h->gk, h->gksize, 1, H, W, st);
    hipLaunchKernelGGL(k_ts_norm, grid1(P), dim3(256), 0, st, h->g, h->gksize ? h->blur : nullptr, h->norm, P);
    hipLaunchKernelGGL(k_ts_masked_sum, dim3(TS_RB), dim3(256), 0, st, h->norm, h->roi_eff, h->partial, P);
    hipLaunchKernelGGL(k_ts_final2, dim3(1), dim3(64), 0, st, h->partial, TS_RB, 2, h->sums);
    hipLaunchKernelGGL(k_ts_scale, grid1(P), dim3(256), 0, st, h->norm, h->sums, h->inorm, P);
    // ---- carrier search on the full spectrum (:454-461)
    if (hipfftSetStream(h->plan, st) != HIPFFT_SUCCESS || hipfftExecR2C(h->plan, (hipfftReal *)h->inorm, (hipfftComplex *)h->F) != HIPFFT_SUCCESS)
        return set_error(VISTAF_E_HIP, "hipfftExecR2C failed");
    hipLaunchKernelGGL(k_ts_mag_full, dim3((W + 255) / 256, H), dim3(256), 0, st, h->F, h->mag, H, W);
    launch_top_peaks(h->mag, 1, H, W, c.seg_dc_exclusion, c.n_peaks, h->peaks, st);
    launch_carrier_choose(h->peaks, c.n_peaks, h->mag, H, W, R, c.seg_peak_max_dy_from_center, h->geom, 1, st);
    CarrierGeom g;
    int cnt_eff = 0;
    TCHK(hipMemcpyAsync(&g, h->geom, sizeof(g), hipMemcpyDeviceToHost, st));
    TCHK(hipMemcpyAsync(&cnt_eff, h->cnt, sizeof(int), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    if (cnt_eff == 0) return set_error(VISTAF_E_STATE, "ROI became empty after saturation exclusion. Lower SAT_THRESH_GRAY / dilation.");
    if (!g.ok) return set_error(VISTAF_E_NOCARRIER, "Could not find FFT peaks for stripe carrier.");
    const int peak_x = g.px_raw, peak_y = g.py_raw;
    if (peak_x - R < 0 || peak_x + R >= W || peak_y - R < 0 || peak_y + R >= H) return set_error(VISTAF_E_NOCARRIER, "carrier band leaves the spectrum");
    // ---- band-pass around the integer peak, in place (:463-468): pruned float64 DFT of the (2R+1)^2 bins, disc window, inverse
    g.x0 = peak_x - R; g.y0 = peak_y - R; g.ph = pm; g.pw = pm; g.dpx = 0.0; g.dpy = 0.0; g.keep_carrier = 1;
    TCHK(hipMemcpyAsync(h->geom, &g, sizeof(g), hipMemcpyHostToDevice, st));
    TCHK(hipStreamSynchronize(st));                                 // `g` is a host temporary
    launch_build_tables(h->geom, 0, h->Ex, h->Ey, h->Gx, h->Gy, 0, 0, 1, H, W, 0, H, W, pm, st);
    launch_dft_forward(h->inorm, nullptr, h->Ex, h->Ey, 0, 0, h->win, h->T, h->patch, pm * pm, 1, H, W, pm, pm, st);
    launch_dft_inverse(h->patch, pm * pm, h->Gx, h->Gy, 0, 0, h->T, h->z, h->amp, nullptr, nullptr, 0, nullptr, nullptr, 1, H, W, pm, pm, st);
    // ---- rotation angle, sign split, which side is dark (:470-491)
    hipLaunchKernelGGL(k_ts_csum, dim3(TS_RB), dim3(256), 0, st, h->z, h->inorm, h->roi_eff, h->partial, P);
    hipLaunchKernelGGL(k_ts_final2, dim3(1), dim3(64), 0, st, h->partial, TS_RB, 2, h->sums);
    double cs[2];
    TCHK(hipMemcpyAsync(cs, h->sums, sizeof(cs), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    const double phi0 = (std::isfinite(cs[0]) && std::isfinite(cs[1])) ? std::atan2(cs[1], cs[0]) : 0.0;
    hipLaunchKernelGGL(k_ts_sign, dim3(TS_RB), dim3(256), 0, st, h->z, std::cos(phi0), std::sin(phi0), h->roi_eff, h->gray, h->ma, h->partial, P);
    hipLaunchKernelGGL(k_ts_final2, dim3(1), dim3(64), 0, st, h->partial, TS_RB, 4, h->sums);
    double ab[4];
    TCHK(hipMemcpyAsync(ab, h->sums, sizeof(ab), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    const double mean_a = ab[1] > 0 ? ab[0] / ab[1] : 1e9, mean_b = ab[3] > 0 ? ab[2] / ab[3] : 1e9;
    const bool a_dark = mean_a <= mean_b;
    const uint8_t *dark_raw = h->ma;
    if (!a_dark) { hipLaunchKernelGGL(k_ts_and_not, grid1(P), dim3(256), 0, st, h->roi_eff, h->ma, h->mb, P); dark_raw = h->mb; }
    // ---- close (kx x ky rectangle) then open, inside roi_eff (:390-406); an empty mask stays empty through all four steps
    const RowSpanSE kc = rect_se(ensure_odd(std::max(1, c.post_close_kx)), ensure_odd(std::max(1, c.post_close_ky)));
    const RowSpanSE ko = rect_se(ensure_odd(std::max(1, c.post_open_kx)), ensure_odd(std::max(1, c.post_open_ky)));
    launch_morph(dark_raw, h->m1, 1, H, W, kc, true, nullptr, nullptr, st, h->prefix);
    launch_morph(h->m1, h->m2, 1, H, W, kc, false, nullptr, nullptr, st, h->prefix);
    launch_morph(h->m2, h->m1, 1, H, W, ko, false, nullptr, nullptr, st, h->prefix);
    launch_morph(h->m1, h->m2, 1, H, W, ko, true, nullptr, h->roi_eff, st, h->prefix);          // dark_final = ... & roi_eff
    hipLaunchKernelGGL(k_ts_and_not, grid1(P), dim3(256), 0, st, h->roi_eff, h->m2, h->m1, P);  // light_final = roi_eff & ~dark_final
    int counts[4] = {0, 0, 0, 0};
    const uint8_t *cm[4] = {d_roi, h->sat, h->m2, h->m1};
    for (int i = 0; i < 4; i++) launch_count_u8(cm[i], h->cnt + i, 1, (int)P, st);
    TCHK(hipMemcpyAsync(counts, h->cnt, sizeof(counts), hipMemcpyDeviceToHost, st));
    if (d_dark) TCHK(hipMemcpyAsync(d_dark, h->m2, P, hipMemcpyDeviceToDevice, st));
    if (d_light) TCHK(hipMemcpyAsync(d_light, h->m1, P, hipMemcpyDeviceToDevice, st));
    if (d_roi_eff) TCHK(hipMemcpyAsync(d_roi_eff, h->roi_eff, P, hipMemcpyDeviceToDevice, st));
    if (d_sat) TCHK(hipMemcpyAsync(d_sat, h->sat, P, hipMemcpyDeviceToDevice, st));
    TCHK(hipStreamSynchronize(st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(VISTAF_E_HIP, std::string("launch: ") + hipGetErrorString(e));
    if (info) {
        for (int i = 0; i < VISTAF_TEMPSEG_NINFO; i++) info[i] = 0.0;
        const double dx = (double)(peak_x - W / 2), dy = (double)(peak_y - H / 2);
        const double fmag = std::hypot(dx / (double)W, dy / (double)H);
        info[VISTAF_TS_PEAK_X] = peak_x; info[VISTAF_TS_PEAK_Y] = peak_y; info[VISTAF_TS_PHI0_RAD] = phi0;
        info[VISTAF_TS_MEAN_GRAY_A] = mean_a; info[VISTAF_TS_MEAN_GRAY_B] = mean_b; info[VISTAF_TS_A_IS_DARK] = a_dark ? 1.0 : 0.0;
        info[VISTAF_TS_ROI_PIXELS] = counts[0]; info[VISTAF_TS_ROI_EFF_PIXELS] = cnt_eff; info[VISTAF_TS_SAT_PIXELS] = counts[1];
        info[VISTAF_TS_DARK_PIXELS] = counts[2]; info[VISTAF_TS_LIGHT_PIXELS] = counts[3];
        info[VISTAF_TS_CARRIER_ANGLE_RAD] = std::atan2(dy, dx);
        info[VISTAF_TS_CARRIER_PERIOD_PX] = fmag > 1e-9 ? 1.0 / fmag : std::nan("");
    }
    return 0;
}

int vistaf_temp_feature_planes(vistaf_tempseg_handle *h, const uint8_t *d_bgr, int blur_ksize, float *d_L, float *d_a, float *d_b, float *d_gray,
                               void *stream)
{
    if (!h || !d_bgr) return set_error(VISTAF_E_INVALID, "null argument");
    const int k = blur_ksize > 1 ? ensure_odd(blur_ksize) : 1;
    if (k != 1 && k != 5) return set_error(VISTAF_E_INVALID, "blur_ksize must be 5 (BLUR_KSIZE as shipped) or <= 1 (no smoothing)");
    launch_feature_planes(d_bgr, h->gamma_tab, h->cbrt_tab, h->lab, k == 5, d_L, d_a, d_b, d_gray, h->H, h->W, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(VISTAF_E_HIP, std::string("launch: ") + hipGetErrorString(e));
    return 0;
}

int vistaf_temp_color_support(vistaf_tempseg_handle *h, const float *d_a, const float *d_b, const uint8_t *d_light, const uint8_t *d_roi_eff,
                              const uint8_t *d_sat, double chroma_min, int dilate_ksize, float *d_chroma, uint8_t *d_support, void *stream)
{
    if (!h || !d_a || !d_b) return set_error(VISTAF_E_INVALID, "null argument");
    if (d_support && (!d_light || !d_roi_eff || !d_sat)) return set_error(VISTAF_E_INVALID, "the support mask needs the light, roi_eff and sat masks");
    hipStream_t st = (hipStream_t)stream;
    const int k = ensure_odd(dilate_ksize);
    if (k > 33) return set_error(VISTAF_E_INVALID, "structuring element taller than 33");
    const uint8_t *light_d = d_light;
    if (d_support && k > 1) { launch_morph(d_light, h->m1, 1, h->H, h->W, ellipse_se(k), true, nullptr, nullptr, st, h->prefix); light_d = h->m1; }
    launch_color_support(d_a, d_b, light_d, d_roi_eff, d_sat, (float)chroma_min, d_chroma, d_support, h->P, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(VISTAF_E_HIP, std::string("launch: ") + hipGetErrorString(e));
    return 0;
}

// ---- map-domain stages (Code/temperature_sensor.py:538-640, :705-747); parity unpinned (k_tempmap.hip)
static int tm_ensure(vistaf_tempseg_handle *h)
{
    if (h->tmA) return 0;
    int rc = 0;
#define TMTRY(x) do { rc = (x); if (rc) return rc; } while (0)
    TMTRY(talloc(h, &h->tmA, h->P)); TMTRY(talloc(h, &h->tmB, h->P));
    TMTRY(talloc(h, &h->tmM1, h->P)); TMTRY(talloc(h, &h->tmM2, h->P));
    TMTRY(talloc(h, &h->tm_kx, (size_t)1024)); TMTRY(talloc(h, &h->tm_ky, (size_t)1024));
    uint8_t *p = nullptr;
    TMTRY(talloc(h, &p, inpaint_scratch_bytes_per_frame(h->H, h->W))); h->tm_scratch = p;
    TMTRY(talloc(h, &h->tm_stats, (size_t)4)); TMTRY(talloc(h, &h->tm_status, (size_t)4)); TMTRY(talloc(h, &h->tm_counts, (size_t)4));
#undef TMTRY
    return 0;
}
static int tm_done(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(VISTAF_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return 0;
}

int vistaf_temp_clamp_map(vistaf_tempseg_handle *h, const float *d_map, const uint8_t *d_roi, double lo, double hi, float *d_out, void *stream)
{
    if (!h || !d_map || !d_roi || !d_out) return set_error(VISTAF_E_INVALID, "null argument");
    launch_tm_clamp(d_map, d_roi, (float)lo, (float)hi, d_out, h->P, (hipStream_t)stream);
    return tm_done("clamp_map");
}

int vistaf_temp_inpaint_map(vistaf_tempseg_handle *h, const float *d_map, const uint8_t *d_roi, int radius, float *d_out, void *stream)
{
    if (!h || !d_map || !d_roi || !d_out) return set_error(VISTAF_E_INVALID, "null argument");
    if (radius < 1 || radius > 100) return set_error(VISTAF_E_INVALID, "inpaint radius out of range");
    int rc = tm_ensure(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    launch_tm_stats(d_map, d_roi, h->tm_stats, h->P, st);
    launch_tm_scale(d_map, d_roi, h->tm_stats, h->tmA, h->tmM1, h->P, st);
    (void)hipMemsetAsync(h->tm_status, 0, sizeof(int32_t), st);
    launch_inpaint_telea(h->tmA, h->tmM1, radius, h->tm_scratch, h->tm_status, nullptr, 1, h->H, h->W, st, true);
    launch_tm_unscale(d_map, d_roi, h->tm_stats, h->tmA, d_out, h->P, st);
    return tm_done("inpaint_temperature_map");
}

int vistaf_temp_fuse_maps(vistaf_tempseg_handle *h, const uint8_t *d_roi, const float *d_wide, const float *d_color, const vistaf_temp_fuse_config *cfg,
                          float *d_final, uint8_t *d_source, int64_t *counts_host, void *stream)
{
    if (!h || !d_roi || !d_wide || !d_color || !d_final || !cfg) return set_error(VISTAF_E_INVALID, "null argument");
    int rc = tm_ensure(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    TmFuse c;
    c.color_lo = (float)(cfg->color_t_min - cfg->color_guard_band); c.color_hi = (float)(cfg->color_t_max + cfg->color_guard_band);
    c.low_th = (float)(cfg->color_t_max - cfg->switch_margin_c); c.high_th = (float)(cfg->color_t_max + cfg->switch_margin_c);
    c.final_lo = (float)cfg->final_t_min; c.final_hi = (float)cfg->final_t_max;
    launch_tm_fuse(d_roi, d_wide, d_color, c, d_final, d_source, counts_host ? h->tm_counts : nullptr, h->P, st);
    if (counts_host) {
        unsigned long long hc[4];
        if (hipMemcpyAsync(hc, h->tm_counts, sizeof(hc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return set_error(VISTAF_E_HIP, "fuse_maps: counters");
        for (int i = 0; i < 4; i++) counts_host[i] = (int64_t)hc[i];
    }
    return tm_done("fuse_maps_per_pixel");
}

static void tm_rotation(double cx, double cy, double angle_deg, double *M)      // cv::getRotationMatrix2D, scale 1
{
    const double a = angle_deg * 3.14159265358979323846 / 180.0, al = std::cos(a), be = std::sin(a);
    M[0] = al; M[1] = be; M[2] = (1 - al) * cx - be * cy;
    M[3] = -be; M[4] = al; M[5] = be * cx + (1 - al) * cy;
}
static TmAff tm_invert(const double *m)                                          // cv::invertAffineTransform
{
    double D = m[0] * m[4] - m[1] * m[3];
    D = D != 0 ? 1.0 / D : 0.0;
    const double A11 = m[4] * D, A22 = m[0] * D, A12 = -m[1] * D, A21 = -m[3] * D;
    TmAff a;
    a.m[0] = A11; a.m[1] = A12; a.m[2] = -A11 * m[2] - A12 * m[5];
    a.m[3] = A21; a.m[4] = A22; a.m[5] = -A21 * m[2] - A22 * m[5];
    return a;
}
static int tm_taps(double sigma, float *d_k, int &n, hipStream_t st)             // cv::getGaussianKernel(ksize(sigma), sigma, CV_32F)
{
    n = ((int)std::nearbyint(sigma * 4 * 2 + 1)) | 1;
    if (n > 1023) return set_error(VISTAF_E_INVALID, "smoothing sigma too large");
    std::vector<double> t(n);
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; t[i] = std::exp(s2 * x * x); sum += t[i]; }
    std::vector<float> f(n);
    for (int i = 0; i < n; i++) f[i] = (float)(t[i] * (1.0 / sum));
    if (hipMemcpyAsync(d_k, f.data(), n * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return set_error(VISTAF_E_HIP, "oriented blur: taps");
    return 0;
}

int vistaf_temp_oriented_blur(vistaf_tempseg_handle *h, const float *d_map, const uint8_t *d_roi, double angle_rad, double sigma_across, double sigma_along,
                              float *d_out, void *stream)
{
    if (!h || !d_map || !d_roi || !d_out) return set_error(VISTAF_E_INVALID, "null argument");
    int rc = tm_ensure(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t P = h->P;
    if (sigma_across <= 0 && sigma_along <= 0) {              // :713-716
        launch_tm_mask_nan(d_map, d_roi, d_out, P, st);
        return tm_done("oriented blur");
    }
    const double cx = h->W / 2.0, cy = h->H / 2.0, angle_deg = -angle_rad * 180.0 / 3.14159265358979323846;
    double M[6], Mi[6];
    tm_rotation(cx, cy, angle_deg, M);
    tm_rotation(cx, cy, -angle_deg, Mi);
    const TmAff a = tm_invert(M), ai = tm_invert(Mi);
    double sx = sigma_across > 0 ? sigma_across : 0.0, sy = sigma_along > 0 ? sigma_along : 0.0;
    if (sx <= 0) return set_error(VISTAF_E_INVALID, "sigma_across must be positive when sigma_along is (cv::GaussianBlur needs sigmaX > 0 for ksize (0, 0))");
    if (sy <= 0) sy = sx;                                     // cv::createGaussianKernels: sigmaY <= 0 takes sigmaX
    int nx = 0, ny = 0;
    if ((rc = tm_taps(sx, h->tm_kx, nx, st)) || (rc = tm_taps(sy, h->tm_ky, ny, st))) return rc;
    launch_tm_zero_nonfinite(d_map, h->tmA, P, st);
    launch_tm_warp_linear(h->tmA, h->tmB, a, h->H, h->W, st);                   // rot_map
    launch_tm_warp_nearest(d_roi, h->tmM1, a, h->H, h->W, st);                  // rot_roi
    launch_gauss_rows(h->tmB, h->tmA, h->tm_kx, nx, 1, h->H, h->W, st);
    launch_gauss_cols(h->tmA, h->tmB, h->tm_ky, ny, 1, h->H, h->W, st);         // blurred
    launch_tm_warp_linear(h->tmB, h->tmA, ai, h->H, h->W, st);                  // back
    launch_tm_warp_nearest(h->tmM1, h->tmM2, ai, h->H, h->W, st);               // back_roi
    launch_tm_mask_nan(h->tmA, h->tmM2, d_out, P, st);
    return tm_done("oriented blur");
}

}  // extern "C"
