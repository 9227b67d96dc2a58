// robust_polyfit2d on the GPU (shape_ftp.py:1086-1136), one 1024-thread workgroup per frame.
//
// IRLS: per iteration a weighted least-squares fit of z on [xn, yn, 1, xn^2, xn*yn, yn^2] (coordinates
// normalised to [-1,1] in float32), then r = z - A.coef, sigma = 1.4826*(median|r - median r| + 1e-6),
// w = 1/(1+(r/(c*sigma))^2).  The reference solves each step with float32 LAPACK lstsq on the N x 6
// system; here the 6x6 normal equations of the same float32-rounded rows (A*w, z*w) are accumulated
// in float64 by a block reduction and solved by Cholesky (condition number of the scaled design is
// O(10), so this is the more accurate of the two).  Medians are exact order statistics (select.hpp).
// Output plane = z - fit (float32, fit evaluated with the reference's operation order), NaN where z is NaN.
#include "kernels.hpp"
#include "select.hpp"

namespace vf {

// compacted fit sample: (z, xn, yn) of every pixel with mask != 0 and finite z (deterministic raster order)
struct FitCtx {
    const float4 *comp;
    int order;
    float coef[6];
    float med;                  // for the MAD pass
    int mode;                   // 0: key = r, 1: key = |r - med|
    __device__ inline float resid(const float4 s) const
    {
        float f = __fmul_rn(coef[0], s.y);
        f = fmaf(coef[1], s.z, f);
        f = __fadd_rn(f, coef[2]);
        if (order >= 2) {
            f = fmaf(coef[3], __fmul_rn(s.y, s.y), f);
            f = fmaf(coef[4], __fmul_rn(s.y, s.z), f);
            f = fmaf(coef[5], __fmul_rn(s.z, s.z), f);
        }
        return __fsub_rn(s.x, f);
    }
    __device__ bool operator()(int e, uint32_t &key) const
    {
        float r = resid(comp[e]);
        if (mode) r = fabsf(__fsub_rn(r, med));
        key = f2key(r);
        return true;
    }
};

// solve the symmetric positive definite n x n system (n <= 6) in place; returns false if not SPD
__device__ inline bool chol_solve(double *A, double *rhs, int n)
{
    double L[36];
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0.0)) return false; L[i * 6 + i] = sqrt(s); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    for (int i = 0; i < n; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * rhs[k]; rhs[i] = s / L[i * 6 + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = rhs[i]; for (int k = i + 1; k < n; k++) s -= L[k * 6 + i] * rhs[k]; rhs[i] = s / L[i * 6 + i]; }
    return true;
}

__global__ __launch_bounds__(SEL_T) void k_robust_polyfit(const float *__restrict__ z_all, const uint8_t *__restrict__ mask_all, int order,
                                                          int iters, float c, int min_count, float *__restrict__ coef_out,
                                                          float *__restrict__ resid_all, float4 *__restrict__ comp_all, int h, int w)
{
    __shared__ SelShared sh;
    __shared__ double s_red[16];
    __shared__ double s_sum[27];
    __shared__ float s_coef[6];
    __shared__ uint32_t s_wcnt[16];
    const size_t b = blockIdx.x;
    const int P = h * w, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nc = order >= 2 ? 6 : 3;
    const float *z = z_all + b * (size_t)P;
    const uint8_t *m = mask_all + b * (size_t)P;
    float4 *comp = comp_all + b * (size_t)P;
    const float cxf = (float)((w - 1) / 2.0), cyf = (float)((h - 1) / 2.0);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // ---- compaction (wave `wid` owns the contiguous pixel range [p0, p1)): (z, xn, yn) of the fitted pixels
    const int Lw = (((P + 15) / 16) + 63) & ~63;
    const int p0 = min(P, wid * Lw), p1 = min(P, p0 + Lw);
    uint32_t cw = 0;
    for (int pb = p0; pb < p1; pb += 64) {
        int p = pb + lane;
        bool ok = p < p1 && m[p] && finitef(z[p]);
        cw += (uint32_t)__popcll(__ballot(ok));
    }
    if (lane == 0) s_wcnt[wid] = cw;
    __syncthreads();
    uint32_t off = 0, n = 0;
    for (int i = 0; i < 16; i++) { uint32_t x = s_wcnt[i]; if (i < wid) off += x; n += x; }
    const bool do_fit = (int)n >= min_count;
    if (do_fit) {
        for (int pb = p0; pb < p1; pb += 64) {
            int p = pb + lane;
            float zz = p < p1 ? z[p] : 0.f;
            bool ok = p < p1 && m[p] && finitef(zz);
            unsigned long long bm = __ballot(ok);
            if (ok) {
                int y = p / w, x = p - y * w;
                float xn = __fdiv_rn(__fsub_rn((float)x, cxf), cxf), yn = __fdiv_rn(__fsub_rn((float)y, cyf), cyf);
                comp[off + (uint32_t)__popcll(bm & lt_mask)] = make_float4(zz, xn, yn, 0.f);
            }
            off += (uint32_t)__popcll(bm);
        }
    }
    __threadfence();
    __syncthreads();

    FitCtx ctx;
    ctx.comp = comp; ctx.order = order; ctx.med = 0.f; ctx.mode = 0;
    for (int i = 0; i < 6; i++) ctx.coef[i] = 0.f;
    float csig = 1.f;      // c * sigma of the previous iteration
    for (int it = 0; do_fit && it < iters; it++) {
        // ---- weighted normal equations over the compacted samples
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
#pragma unroll 2
        for (int e = tid; e < (int)n; e += SEL_T) {
            const float4 s = comp[e];
            float wt = 1.f;
            if (it > 0) {
                float u = __fdiv_rn(ctx.resid(s), csig);
                wt = __fdiv_rn(1.0f, __fadd_rn(1.0f, __fmul_rn(u, u)));
            }
            float a[6];
            a[0] = __fmul_rn(s.y, wt); a[1] = __fmul_rn(s.z, wt); a[2] = wt;
            a[3] = __fmul_rn(__fmul_rn(s.y, s.y), wt); a[4] = __fmul_rn(__fmul_rn(s.y, s.z), wt); a[5] = __fmul_rn(__fmul_rn(s.z, s.z), wt);
            double zw = (double)__fmul_rn(s.x, wt);
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = i; j < 6; j++) { acc[k] = fma((double)a[i], (double)a[j], acc[k]); k++; }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) acc[21 + i] = fma((double)a[i], zw, acc[21 + i]);
        }
        for (int i = 0; i < 27; i++) {
            double v = block_sum<double>(acc[i], s_red);
            if (tid == 0) s_sum[i] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double A[36], rhs[6];
            int k = 0;
            for (int i = 0; i < 6; i++)
                for (int j = i; j < 6; j++) { A[i * 6 + j] = s_sum[k]; A[j * 6 + i] = s_sum[k]; k++; }
            for (int i = 0; i < 6; i++) rhs[i] = s_sum[21 + i];
            bool ok = chol_solve(A, rhs, nc);
            for (int i = 0; i < 6; i++) s_coef[i] = (ok && i < nc) ? (float)rhs[i] : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < 6; i++) ctx.coef[i] = s_coef[i];
        if (it == iters - 1) break;   // the weights of the last iteration are never used upstream
        // ---- sigma = 1.4826 * (median |r - median r| + 1e-6)
        uint32_t nn, kmin, kmax;
        ctx.mode = 0;
        block_minmax(ctx, (int)n, sh, nn, kmin, kmax);
        float medr = block_median(ctx, (int)n, sh, nn, kmin, kmax);
        __syncthreads();
        ctx.med = medr; ctx.mode = 1;
        block_minmax(ctx, (int)n, sh, nn, kmin, kmax);
        float mad = block_median(ctx, (int)n, sh, nn, kmin, kmax);
        __syncthreads();
        ctx.mode = 0;
        mad = __fadd_rn(mad, 1e-6f);
        float sigma = __fmul_rn(1.4826f, mad);
        csig = __fmul_rn(c, sigma);
    }
    if (tid < 6) coef_out[b * 6 + tid] = do_fit ? ctx.coef[tid] : 0.f;
    // residual plane: z - fit, fit evaluated as eval_poly2d does (shape_ftp.py:1093-1097, :1132-1135)
    float *out = resid_all + b * (size_t)P;
    for (int p = tid; p < P; p += SEL_T) {
        float zz = z[p];
        float fit = 0.f;
        if (do_fit) {
            int y = p / w, x = p - y * w;
            float xn = __fdiv_rn(__fsub_rn((float)x, cxf), cxf), yn = __fdiv_rn(__fsub_rn((float)y, cyf), cyf);
            fit = __fadd_rn(__fadd_rn(__fmul_rn(ctx.coef[0], xn), __fmul_rn(ctx.coef[1], yn)), ctx.coef[2]);
            if (order >= 2) {
                fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(ctx.coef[3], xn), xn));
                fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(ctx.coef[4], xn), yn));
                fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(ctx.coef[5], yn), yn));
            }
        }
        out[p] = __fsub_rn(zz, fit);
    }
}

// min_count: 200 upstream (:1103); 500 for the debug_ramp call (shape_ftp.py:1365)
void launch_robust_polyfit(const float *z, const uint8_t *mask, int order, int iters, float c, int min_count, float *coef_out,
                           float *resid_out, void *comp_scratch, int B, int h, int w, hipStream_t st)
{
    hipLaunchKernelGGL(k_robust_polyfit, dim3(B), dim3(SEL_T), 0, st, z, mask, order, iters, c, min_count, coef_out, resid_out,
                       (float4 *)comp_scratch, h, w);
}

}  // namespace vf
