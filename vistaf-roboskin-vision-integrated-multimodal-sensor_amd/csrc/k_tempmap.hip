// Map-domain stages of the temperature modality (Code/temperature_sensor.py:538-640, :705-747): clamp_map, inpaint_temperature_map,
// fuse_maps_per_pixel, oriented_gaussian_blur_float.  PARITY UNPINNED: the reference tree holds no output of these stages (its
// temperature_map_*.npy are among the blobs that were not mounted) and the regressors that feed them only exist as pickles; the kernels are
// checked on synthetic planes against the test suite's CPU restatement of the source text (temp_oracle.py).  All streaming, one thread per pixel.
#include "kernels.hpp"

namespace vf {

namespace {
constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS;
__device__ inline long long tm_cvr(double v) { return __double2ll_rn(v); }
__device__ inline void tm_src_coord(const TmAff &a, int x, int y, int round_delta, int shift, int &X, int &Y)
{
    const long long ad = tm_cvr(a.m[0] * x * AB_SCALE), bd = tm_cvr(a.m[3] * x * AB_SCALE);
    const long long X0 = tm_cvr((a.m[1] * y + a.m[2]) * AB_SCALE) + round_delta, Y0 = tm_cvr((a.m[4] * y + a.m[5]) * AB_SCALE) + round_delta;
    X = (int)((X0 + ad) >> shift);
    Y = (int)((Y0 + bd) >> shift);
}
__device__ inline int tm_reflect(int p, int n)       // BORDER_REFLECT
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}
}  // namespace

// clamp_map (:538-543): clipped inside the ROI where finite, NaN outside
__global__ void k_tm_clamp(const float *__restrict__ m, const uint8_t *__restrict__ roi, float lo, float hi, float *__restrict__ out, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float v = m[i];
    if (!roi[i]) v = __uint_as_float(0x7fc00000u);
    else if (finitef(v)) v = fminf(fmaxf(v, lo), hi);
    out[i] = v;
}

// inpaint_temperature_map (:546-580), step 1: counts and range of the known pixels.  stats: [0] known, [1] missing, [2] min key, [3] max key
__global__ void k_tm_stats(const float *__restrict__ m, const uint8_t *__restrict__ roi, uint32_t *__restrict__ stats, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < P && roi[i];
    const float v = in ? m[i] : 0.f;
    const bool known = in && finitef(v), missing = in && !finitef(v);
    const unsigned long long bk = __ballot(known), bm = __ballot(missing);
    uint32_t kmin = known ? f2key(v) : 0xFFFFFFFFu, kmax = known ? f2key(v) : 0u;
    for (int o = 32; o; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)kmin, o, 64), b = (uint32_t)__shfl_xor((int)kmax, o, 64);
        kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
    }
    if ((threadIdx.x & 63) == 0) {
        if (bk) { atomicAdd(&stats[0], (uint32_t)__popcll(bk)); atomicMin(&stats[2], kmin); atomicMax(&stats[3], kmax); }
        if (bm) atomicAdd(&stats[1], (uint32_t)__popcll(bm));
    }
}
// step 2: the 8-bit image (as floats holding 0..255) and the mask of the missing pixels
__global__ void k_tm_scale(const float *__restrict__ m, const uint8_t *__restrict__ roi, const uint32_t *__restrict__ stats, float *__restrict__ scaled,
                           uint8_t *__restrict__ miss, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float vmin = key2f(stats[2]), vmax = key2f(stats[3]);
    const float d = (float)((double)vmax - (double)vmin);                 // Python double difference, a float32 scalar inside the array expression
    const bool in = roi[i] != 0;
    const float v = m[i];
    float s = 0.f;
    if (in && finitef(v)) {
        float t = __fmul_rn(__fdiv_rn(__fsub_rn(v, vmin), d), 255.0f);
        t = fminf(fmaxf(t, 0.f), 255.f);
        s = (float)(uint8_t)t;                                           // astype(np.uint8): truncation
    }
    scaled[i] = s;
    miss[i] = (in && !finitef(v) && stats[0] && ((double)vmax - (double)vmin >= 1e-6)) ? 1 : 0;        // nothing to march in the two early-return cases
}
// step 3: back to temperatures; the two early returns of the reference (nothing missing / nothing known; flat map) are decided per frame here
__global__ void k_tm_unscale(const float *__restrict__ m, const uint8_t *__restrict__ roi, const uint32_t *__restrict__ stats, const float *__restrict__ filled,
                             float *__restrict__ out, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float qnan = __uint_as_float(0x7fc00000u);
    if (!roi[i]) { out[i] = qnan; return; }
    const float v = m[i];
    if (stats[0] == 0 || stats[1] == 0) { out[i] = v; return; }
    const float vmin = key2f(stats[2]), vmax = key2f(stats[3]);
    const double dd = (double)vmax - (double)vmin;
    if (dd < 1e-6) { out[i] = finitef(v) ? v : vmin; return; }
    out[i] = __fadd_rn(__fmul_rn(__fdiv_rn(filled[i], 255.0f), (float)dd), vmin);
}

// fuse_maps_per_pixel (:594-636); counts: roi, wide_ok, color_ok, blend
__global__ void k_tm_fuse(const uint8_t *__restrict__ roi, const float *__restrict__ wide, const float *__restrict__ color, TmFuse c, float *__restrict__ fin,
                          uint8_t *__restrict__ source, unsigned long long *__restrict__ counts, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < P;
    const bool r = act && roi[i];
    const float wv = act ? wide[i] : 0.f, cv = act ? color[i] : 0.f;
    const bool wide_ok = r && finitef(wv);
    const bool color_ok = r && finitef(cv) && cv >= c.color_lo && cv <= c.color_hi;
    const bool blend = wide_ok && color_ok && wv > c.low_th && wv < c.high_th;
    if (act) {
        float f = color_ok ? cv : wv;
        uint8_t s = color_ok ? 255 : 0;
        if (blend) {
            float wgt = __fdiv_rn(__fsub_rn(c.high_th, wv), __fsub_rn(c.high_th, c.low_th));
            wgt = fminf(fmaxf(wgt, 0.f), 1.f);
            f = __fadd_rn(__fmul_rn(wgt, cv), __fmul_rn(__fsub_rn(1.0f, wgt), wv));
            s = 128;
        }
        if (!r) f = __uint_as_float(0x7fc00000u);
        else if (finitef(f)) f = fminf(fmaxf(f, c.final_lo), c.final_hi);
        fin[i] = f;
        if (source) source[i] = s;
    }
    const unsigned long long b0 = __ballot(r), b1 = __ballot(wide_ok), b2 = __ballot(color_ok), b3 = __ballot(blend);
    if (counts && (threadIdx.x & 63) == 0) {
        if (b0) atomicAdd(&counts[0], (unsigned long long)__popcll(b0));
        if (b1) atomicAdd(&counts[1], (unsigned long long)__popcll(b1));
        if (b2) atomicAdd(&counts[2], (unsigned long long)__popcll(b2));
        if (b3) atomicAdd(&counts[3], (unsigned long long)__popcll(b3));
    }
}

// oriented_gaussian_blur_float (:705-747) pieces
__global__ void k_tm_zero_nonfinite(const float *__restrict__ m, float *__restrict__ out, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) { const float v = m[i]; out[i] = finitef(v) ? v : 0.f; }
}
// warpAffine(src f32, INTER_LINEAR, BORDER_REFLECT) with the inverse map `a`: fixed-point source coordinates (1/32 pixel), float weights
__global__ void k_tm_warp_linear(const float *__restrict__ src, float *__restrict__ dst, TmAff a, int h, int w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    int X, Y;
    tm_src_coord(a, x, y, AB_SCALE / INTER_TAB / 2, AB_BITS - INTER_BITS, X, Y);
    const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS, ax = X & (INTER_TAB - 1), ay = Y & (INTER_TAB - 1);
    const int xa = tm_reflect(sx, w), xb = tm_reflect(sx + 1, w), ya = tm_reflect(sy, h), yb = tm_reflect(sy + 1, h);
    const float fx = __fdiv_rn((float)ax, (float)INTER_TAB), fy = __fdiv_rn((float)ay, (float)INTER_TAB);
    const float gx = __fsub_rn(1.0f, fx), gy = __fsub_rn(1.0f, fy);
    const float w00 = __fmul_rn(gx, gy), w01 = __fmul_rn(fx, gy), w10 = __fmul_rn(gx, fy), w11 = __fmul_rn(fx, fy);
    const float p00 = src[(size_t)ya * w + xa], p01 = src[(size_t)ya * w + xb], p10 = src[(size_t)yb * w + xa], p11 = src[(size_t)yb * w + xb];
    dst[(size_t)y * w + x] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(p00, w00), __fmul_rn(p01, w01)), __fmul_rn(p10, w10)), __fmul_rn(p11, w11));
}
// warpAffine(mask u8 0 / 1, INTER_NEAREST, BORDER_CONSTANT 0)
__global__ void k_tm_warp_nearest(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, TmAff a, int h, int w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    int X, Y;
    tm_src_coord(a, x, y, AB_SCALE / 2, AB_BITS, X, Y);
    dst[(size_t)y * w + x] = (X >= 0 && X < w && Y >= 0 && Y < h && src[(size_t)Y * w + X]) ? 1 : 0;
}
__global__ void k_tm_mask_nan(const float *__restrict__ m, const uint8_t *__restrict__ keep, float *__restrict__ out, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) out[i] = keep[i] ? m[i] : __uint_as_float(0x7fc00000u);
}

static inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

void launch_tm_clamp(const float *m, const uint8_t *roi, float lo, float hi, float *out, size_t P, hipStream_t st)
{
    hipLaunchKernelGGL(k_tm_clamp, g1(P), dim3(256), 0, st, m, roi, lo, hi, out, P);
}
void launch_tm_stats(const float *m, const uint8_t *roi, uint32_t *stats, size_t P, hipStream_t st)
{
    static const uint32_t init[4] = {0u, 0u, 0xFFFFFFFFu, 0u};
    (void)hipMemcpyAsync(stats, init, sizeof(init), hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_tm_stats, g1(P), dim3(256), 0, st, m, roi, stats, P);
}
void launch_tm_scale(const float *m, const uint8_t *roi, const uint32_t *stats, float *scaled, uint8_t *miss, size_t P, hipStream_t st)
{
    hipLaunchKernelGGL(k_tm_scale, g1(P), dim3(256), 0, st, m, roi, stats, scaled, miss, P);
}
void launch_tm_unscale(const float *m, const uint8_t *roi, const uint32_t *stats, const float *filled, float *out, size_t P, hipStream_t st)
{
    hipLaunchKernelGGL(k_tm_unscale, g1(P), dim3(256), 0, st, m, roi, stats, filled, out, P);
}
void launch_tm_fuse(const uint8_t *roi, const float *wide, const float *color, const TmFuse &c, float *fin, uint8_t *source, unsigned long long *counts, size_t P,
                    hipStream_t st)
{
    if (counts) (void)hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), st);
    hipLaunchKernelGGL(k_tm_fuse, g1(P), dim3(256), 0, st, roi, wide, color, c, fin, source, counts, P);
}
void launch_tm_zero_nonfinite(const float *m, float *out, size_t P, hipStream_t st) { hipLaunchKernelGGL(k_tm_zero_nonfinite, g1(P), dim3(256), 0, st, m, out, P); }
void launch_tm_warp_linear(const float *src, float *dst, const TmAff &a, int h, int w, hipStream_t st)
{
    hipLaunchKernelGGL(k_tm_warp_linear, dim3((w + 255) / 256, h), dim3(256), 0, st, src, dst, a, h, w);
}
void launch_tm_warp_nearest(const uint8_t *src, uint8_t *dst, const TmAff &a, int h, int w, hipStream_t st)
{
    hipLaunchKernelGGL(k_tm_warp_nearest, dim3((w + 255) / 256, h), dim3(256), 0, st, src, dst, a, h, w);
}
void launch_tm_mask_nan(const float *m, const uint8_t *keep, float *out, size_t P, hipStream_t st) { hipLaunchKernelGGL(k_tm_mask_nan, g1(P), dim3(256), 0, st, m, keep, out, P); }

}  // namespace vf
