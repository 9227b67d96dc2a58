// robust_polyfit2d on the GPU (shape_ftp.py:1086-1136), one 1024-thread workgroup per frame.
//
// IRLS: per iteration a weighted least-squares fit of z on [xn, yn, 1, xn^2, xn*yn, yn^2] (coordinates
// normalised to [-1,1] in float32), then r = z - A.coef, sigma = 1.4826*(median|r - median r| + 1e-6),
// w = 1/(1+(r/(c*sigma))^2).  The reference solves each step with float32 LAPACK lstsq on the N x 6
// system; here the 6x6 normal equations of the same float32-rounded rows (A*w, z*w) are accumulated
// in float64 by a block reduction and solved by Cholesky (condition number of the scaled design is
// O(10), so this is the more accurate of the two).  Medians are exact order statistics (select.hpp).
// Output plane = z - fit (float32, fit evaluated with the reference's operation order), NaN where z is NaN.
//
// Every pass walks the plane itself (z f32 + mask u8 = 5 B/px, coordinates from two LDS tables) with
// SEL_U / FIT_U independent loads in flight per thread: the passes are bound by memory round trips, not bytes.
#include <cstdio>
#include <algorithm>
#include <type_traits>
#include "kernels.hpp"
#ifdef VISTAF_DEBUG
// cycle sums of frame 0's fits (thread 0): [0] histogram clear, [1] histogram pass, [2] bucket search, [3] candidate collection, [4] sort,
// [5] second order statistic by a pass, [6] column sums, [7] reduction + solve, [8] load, [9] residual plane, [10] launches
namespace vf { __device__ unsigned long long g_fit_dbg[16]; __device__ unsigned long long g_fit_last; }
#define SEL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); vf::g_fit_dbg[i] += t_ - vf::g_fit_last; vf::g_fit_last = t_; } } while (0)
#define FIT_START() do { if (blockIdx.x == 0 && threadIdx.x == 0) { vf::g_fit_last = __builtin_amdgcn_s_memtime(); vf::g_fit_dbg[10]++; } } while (0)
#else
#define FIT_START() do { } while (0)
#endif
#include "select.hpp"

namespace vf {

constexpr int FIT_TAB = 4096;     // LDS coordinate tables: xn[w] followed by yn[h] when w + h <= FIT_TAB
constexpr int FIT_YTAB = 1280;    // rows of the column variant's yn table (5 KB: the kernel stays at 44 KB of LDS, see k_inpaint_win.hip's first tier)
constexpr int FIT_U = 8;          // samples per thread in flight in the normal-equation pass

struct FitCtx {
    const float *z;
    const uint8_t *m;
    const float *tab;           // LDS: xn[0..w), yn[0..h)
    bool use_tab;               // false: divide on the fly (tables too small)
    int w;
    uint32_t magic;             // i / w == umulhi(i, magic) (0: plain division; chosen on the host)
    float cxf, cyf;
    int order;
    float coef[6];
    float med;                  // for the MAD pass
    int mode;                   // 0: key = r, 1: key = |r - med|
    __device__ inline void coords(int i, float &xn, float &yn) const
    {
        int y = magic ? (int)__umulhi((uint32_t)i, magic) : i / w, x = i - y * w;
        if (use_tab) { xn = tab[x]; yn = tab[w + y]; }
        else { xn = __fdiv_rn(__fsub_rn((float)x, cxf), cxf); yn = __fdiv_rn(__fsub_rn((float)y, cyf), cyf); }
    }
    __device__ inline float resid(float zz, float xn, float yn) const
    {
        float f = __fmul_rn(coef[0], xn);
        f = fmaf(coef[1], yn, f);
        f = __fadd_rn(f, coef[2]);
        if (order >= 2) {
            f = fmaf(coef[3], __fmul_rn(xn, xn), f);
            f = fmaf(coef[4], __fmul_rn(xn, yn), f);
            f = fmaf(coef[5], __fmul_rn(yn, yn), f);
        }
        return __fsub_rn(zz, f);
    }
    // z is the kernel's working copy of the plane: NaN wherever the pixel is not fitted (mask 0 or non-finite input)
    __device__ inline bool sample(int i, float &zz, float &xn, float &yn) const
    {
        zz = z[i];
        coords(i, xn, yn);
        return finitef(zz);
    }
    __device__ bool operator()(int i, uint32_t &key) const
    {
        float zz, xn, yn;
        bool ok = sample(i, zz, xn, yn);
        float r = resid(zz, xn, yn);
        if (mode) r = fabsf(__fsub_rn(r, med));
        key = f2key(r);
        return ok;
    }
};

// solve the symmetric positive definite N x N system in place (fully unrolled: everything stays in registers);
// returns false if not SPD
template <int N>
__device__ inline bool chol_solve(double (&A)[6][6], double (&rhs)[6])
{
    // one reciprocal per pivot (the 27 divisions of the textbook form are a long dependent chain on the one thread that solves)
    double L[N][N], inv[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) { if (!(s > 0.0)) return false; L[i][i] = sqrt(s); inv[i] = 1.0 / L[i][i]; }
            else L[i][j] = s * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = rhs[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i][k] * rhs[k];
        rhs[i] = s * inv[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = rhs[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= L[k][i] * rhs[k];
        rhs[i] = s * inv[i];
    }
    return true;
}

__global__ __launch_bounds__(SEL_T) void k_robust_polyfit(const float *__restrict__ z_all, const uint8_t *__restrict__ mask_all, int order,
                                                          int iters, float c, int min_count, int min_mask_count, float *__restrict__ coef_out,
                                                          float *__restrict__ resid_all, int h, int w, uint32_t magic)
{
    __shared__ SelShared sh;
    __shared__ double s_part[16][27];
    __shared__ double s_sum[27];
    __shared__ float s_coef[6];
    __shared__ float s_tab[FIT_TAB];
    const size_t b = blockIdx.x;
    const int P = h * w, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nc = order >= 2 ? 6 : 3;
    const float *z = z_all + b * (size_t)P;
    const uint8_t *m = mask_all + b * (size_t)P;
    const float cxf = (float)((w - 1) / 2.0), cyf = (float)((h - 1) / 2.0);
    const bool use_tab = w + h <= FIT_TAB;
    if (use_tab) {
        for (int i = tid; i < w + h; i += SEL_T)
            s_tab[i] = i < w ? __fdiv_rn(__fsub_rn((float)i, cxf), cxf) : __fdiv_rn(__fsub_rn((float)(i - w), cyf), cyf);
    }
    __syncthreads();

    // Working copy in the output plane (it is only written at the very end): z where the pixel is fitted, NaN elsewhere.  The ~40 passes
    // of the fit then read 4 B per pixel instead of 5 (they run at HBM / Infinity-Cache speed: 256 frames do not fit the L2s).  Every
    // thread only ever touches the pixels tid, tid + 1024, ...: its own stores, so no fence is needed.
    float *zc = resid_all + b * (size_t)P;
    // the same pass counts the fitted pixels and finds the range of z (coefficients are still zero: r = z)
    uint32_t cnt0 = 0, cntm = 0;
    unsigned long long mn0 = ~0ull, mx0 = 0;
    {
        int p = tid;
        for (; p + (FIT_U - 1) * SEL_T < P; p += FIT_U * SEL_T) {
            float zz[FIT_U];
            uint8_t mk[FIT_U];
#pragma unroll
            for (int u = 0; u < FIT_U; u++) { zz[u] = z[p + u * SEL_T]; mk[u] = m[p + u * SEL_T]; }
#pragma unroll
            for (int u = 0; u < FIT_U; u++) {
                const bool ok = mk[u] && finitef(zz[u]);
                cntm += mk[u] != 0;
                zc[p + u * SEL_T] = ok ? zz[u] : __uint_as_float(0x7fc00000u);
                if (ok) { const uint32_t key = f2key(zz[u]); cnt0++; if (key < mn0) mn0 = key; if (key + 1ull > mx0) mx0 = key + 1ull; }
            }
        }
        for (; p < P; p += SEL_T) {
            const float zz = z[p];
            const bool ok = m[p] && finitef(zz);
            cntm += m[p] != 0;
            zc[p] = ok ? zz : __uint_as_float(0x7fc00000u);
            if (ok) { const uint32_t key = f2key(zz); cnt0++; if (key < mn0) mn0 = key; if (key + 1ull > mx0) mx0 = key + 1ull; }
        }
    }
    FitCtx ctx;
    ctx.z = zc; ctx.m = m; ctx.tab = s_tab; ctx.use_tab = use_tab; ctx.w = w; ctx.magic = magic; ctx.cxf = cxf; ctx.cyf = cyf;
    ctx.order = order; ctx.med = 0.f; ctx.mode = 0;
    for (int i = 0; i < 6; i++) ctx.coef[i] = 0.f;

    // ---- number of fitted pixels (mask != 0 and finite z) and the range of z (coefficients are still zero: r = z)
    uint32_t n, zkmin, zkmax;
    {
        __syncthreads();
        n = block_sum<uint32_t>(cnt0, sh.wsum);
        if (min_mask_count > 0) cntm = block_sum<uint32_t>(cntm, sh.wsum);
        mn0 = block_min_u64(mn0, sh.red64);
        mx0 = block_max_u64(mx0, sh.red64);
        zkmin = (uint32_t)mn0; zkmax = mx0 ? (uint32_t)(mx0 - 1) : 0;
        __syncthreads();
    }
    const float zmin = key2f(zkmin), zmax = key2f(zkmax);
    const bool do_fit = (int)n >= min_count && (min_mask_count <= 0 || (int)cntm >= min_mask_count);

    float csig = 1.f;      // c * sigma of the previous iteration
    for (int it = 0; do_fit && it < iters; it++) {
        // ---- weighted normal equations
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
        auto accumulate = [&](float zz, float xn, float yn) {
            float wt = 1.f;
            if (it > 0) {
                float u = __fdiv_rn(ctx.resid(zz, xn, yn), csig);
                wt = __fdiv_rn(1.0f, __fadd_rn(1.0f, __fmul_rn(u, u)));
            }
            float a[6];
            a[0] = __fmul_rn(xn, wt); a[1] = __fmul_rn(yn, wt); a[2] = wt;
            a[3] = __fmul_rn(__fmul_rn(xn, xn), wt); a[4] = __fmul_rn(__fmul_rn(xn, yn), wt); a[5] = __fmul_rn(__fmul_rn(yn, yn), wt);
            double zw = (double)__fmul_rn(zz, wt);
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = i; j < 6; j++) { acc[k] = fma((double)a[i], (double)a[j], acc[k]); k++; }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) acc[21 + i] = fma((double)a[i], zw, acc[21 + i]);
        };
        {
            int i = tid;
            for (; i + (FIT_U - 1) * SEL_T < P; i += FIT_U * SEL_T) {
                float zz[FIT_U], xn[FIT_U], yn[FIT_U];
                bool ok[FIT_U];
#pragma unroll
                for (int u = 0; u < FIT_U; u++) ok[u] = ctx.sample(i + u * SEL_T, zz[u], xn[u], yn[u]);
#pragma unroll
                for (int u = 0; u < FIT_U; u++)
                    if (ok[u]) accumulate(zz[u], xn[u], yn[u]);
            }
            for (; i < P; i += SEL_T) {
                float zz, xn, yn;
                if (ctx.sample(i, zz, xn, yn)) accumulate(zz, xn, yn);
            }
        }
        // 27 sums: DPP network inside each wave, then the 16 wave partials in a fixed order
#pragma unroll
        for (int i = 0; i < 27; i++) {
            double v = wave_sum(acc[i]);
            if (lane == 0) s_part[wid][i] = v;
        }
        __syncthreads();
        if (tid < 27) {
            double v = 0.0;
            for (int k = 0; k < 16; k++) v += s_part[k][tid];
            s_sum[tid] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double A[6][6], rhs[6];
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = i; j < 6; j++) { A[i][j] = s_sum[k]; A[j][i] = s_sum[k]; k++; }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) rhs[i] = s_sum[21 + i];
            bool ok = nc == 6 ? chol_solve<6>(A, rhs) : chol_solve<3>(A, rhs);
#pragma unroll
            for (int i = 0; i < 6; i++) s_coef[i] = (ok && i < nc) ? (float)rhs[i] : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < 6; i++) ctx.coef[i] = s_coef[i];
        if (it == iters - 1) break;   // the weights of the last iteration are never used upstream
        // ---- sigma = 1.4826 * (median |r - median r| + 1e-6)
        // range of r = z - fit without a pass over the data: |xn|, |yn| <= 1, so |fit| <= sum |coef| (plus rounding slack); the
        // histogram refinement only needs a range that CONTAINS the values
        uint32_t nn = n, kmin, kmax;
        ctx.mode = 0;
        {
            float fb = 0.f;
            for (int i = 0; i < 6; i++) fb += fabsf(ctx.coef[i]);
            fb = fb * 1.0001f + 1e-30f;
            kmin = f2key(zmin - fb - 1e-6f * fabsf(zmin)); kmax = f2key(zmax + fb + 1e-6f * fabsf(zmax));
        }
        float medr = block_median(ctx, P, sh, nn, kmin, kmax);
        __syncthreads();
        ctx.med = medr; ctx.mode = 1;
        {
            // |r - med| lies in [0, max(rmax - med, med - rmin)] (float subtraction is monotone): no second min/max pass
            float hi1 = fabsf(__fsub_rn(key2f(kmax), medr)), hi2 = fabsf(__fsub_rn(key2f(kmin), medr));
            kmin = f2key(0.f); kmax = f2key(hi1 > hi2 ? hi1 : hi2);
        }
        float mad = block_median(ctx, P, sh, nn, kmin, kmax);
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
            float xn, yn;
            ctx.coords(p, xn, yn);
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

// ------------------------------------------------------------------------------------------------------------------------------------
// Register-resident, column-owning variant for the batched frame sizes (w <= 1024, h / floor(1024 / w64) <= 64; 224 x 224: RP = 56).
//
// Thread t owns ONE image column x (threads run along x, padded to a multiple of 64 so that every wave lies inside one row group) and
// the rows y = grp + groups * k, k < RP, of it; its samples stay in RP VGPRs for the whole fit (NaN = not fitted).  The ~35 passes of an
// IRLS fit then cost neither memory traffic (the first variants walked a working copy of the plane: 5.3 GB of HBM traffic per batch of
// 256 for 0.35 GB of input and output) nor coordinate arithmetic: with x fixed per thread and y uniform per wave,
//     fit(x, y) = (c0 x + c2 + c3 x^2) + y (c1 + c4 x) + c5 y^2 = A_t + y B_t + C_y        -- one fma and one add per sample,
// and the normal equations are accumulated as the 8 per-thread column sums  P_b = sum_k w^2 y^b (b = 0..4),  Q_b = sum_k w^2 z y^b
// (b = 0..2) in float64, multiplied by the powers of x once per iteration and thread:  (A^T W^2 A)_ij = sum x^a y^b w^2  over the 15
// monomials of degree <= 4, rhs_i likewise.  Same minimisation as the reference's lstsq on (A * w, z * w) (:1119-1121); float64 sums of
// exact products instead of float32-rounded rows, i.e. closer to the exact solution than either LAPACK's float32 SVD or the first variant.
// Medians are exact order statistics as before.  The residual plane is evaluated with eval_poly2d's own operation order (:1093-1097).
// GT: row groups as a compile-time constant (0: the runtime value).  With a constant stride every yn read is one LDS read at an immediate
// offset from a single base; with a runtime stride the RP row addresses of a pass are scalar values of their own (hundreds of SGPR spills).
template <int RP, int NT, int GT>
__device__ __attribute__((always_inline)) inline void robust_polyfit_col_body(const float *__restrict__ z_all, const uint8_t *__restrict__ mask_all, int order,
                                                              int iters, float c, int min_count, int min_mask_count, float *__restrict__ coef_out,
                                                              float *__restrict__ resid_all, int h, int w, int cols_pad, int groups_rt)
{
    // everything below that depends on the workgroup size takes it from NT (block reductions through 16-entry scratch arrays, NT / 64
    // partial sums, the selection's NT-strided loops): the launcher must start exactly NT threads and derive the row groups from NT
    static_assert(NT % 64 == 0 && NT >= 256 && NT / 64 <= 16, "block_sum / block_min / block_max scratch holds 16 waves");
    static_assert(RP >= 1 && RP <= 128, "rows per thread live in registers");
    static_assert(GT >= 0 && (GT == 0 || (RP - 1) * GT + GT + 15 < FIT_YTAB), "row table");
    const int groups = GT > 0 ? GT : groups_rt;
    FIT_START();
    __shared__ SelShared sh;
    __shared__ double s_part[NT / 64][21];
    __shared__ double s_sum[21];
    __shared__ float s_coef[6];
    __shared__ float s_yn[FIT_YTAB];
    const size_t b = blockIdx.x;
    const int P = h * w, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nc = order >= 2 ? 6 : 3;
    const float *z = z_all + b * (size_t)P;
    const uint8_t *m = mask_all + b * (size_t)P;
    const float cxf = (float)((w - 1) / 2.0), cyf = (float)((h - 1) / 2.0);
    for (int i = tid; i < FIT_YTAB; i += NT) s_yn[i] = __fdiv_rn(__fsub_rn((float)i, cyf), cyf);       // rows >= h: only read for samples that are NaN
    const int col = tid % cols_pad, grp = __builtin_amdgcn_readfirstlane(tid / cols_pad);   // uniform inside a wave (cols_pad % 64 == 0)
    const bool owner = col < w && grp < groups;
    const float xn = __fdiv_rn(__fsub_rn((float)col, cxf), cxf);
    const float qnan = __uint_as_float(0x7fc00000u);
    // ---- load this thread's column samples; count fitted / masked pixels, range of z
    float zr[RP];
    uint32_t cnt0 = 0, cntm = 0;
    unsigned long long mn0 = ~0ull, mx0 = 0;
    {
        constexpr int LU = 8;
        uint8_t mk[LU];
#pragma unroll
        for (int u0 = 0; u0 < RP; u0 += LU) {
#pragma unroll
            for (int v = 0; v < LU; v++) {
                const int u = u0 + v;
                if (u < RP) {
                    const int y = grp + groups * u;
                    const bool in = owner && y < h;
                    zr[u] = in ? z[(size_t)y * w + col] : qnan;
                    mk[v] = in ? m[(size_t)y * w + col] : (uint8_t)0;
                }
            }
#pragma unroll
            for (int v = 0; v < LU; v++) {
                const int u = u0 + v;
                if (u < RP) {
                    const float zz = zr[u];
                    const bool ok = mk[v] && finitef(zz);
                    cntm += mk[v] != 0;
                    zr[u] = ok ? zz : qnan;
                    if (ok) { const uint32_t key = f2key(zz); cnt0++; if (key < mn0) mn0 = key; if (key + 1ull > mx0) mx0 = key + 1ull; }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t n = block_sum<uint32_t>(cnt0, sh.wsum);
    if (min_mask_count > 0) cntm = block_sum<uint32_t>(cntm, sh.wsum);
    mn0 = block_min_u64(mn0, sh.red64);
    mx0 = block_max_u64(mx0, sh.red64);
    const uint32_t zkmin = (uint32_t)mn0, zkmax = mx0 ? (uint32_t)(mx0 - 1) : 0;
    __syncthreads();
    const float zmin = key2f(zkmin), zmax = key2f(zkmax);
    const bool do_fit = (int)n >= min_count && (min_mask_count <= 0 || (int)cntm >= min_mask_count);
    SEL_STAMP(8);

    float coef[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float At = 0.f, Bt = 0.f;          // fit(x, y) = At + yn * Bt + c5 * yn^2 for this thread's column
    float med = 0.f;
    int mode = 0;                      // 0: key = r, 1: key = |r - med|
    // residual of sample u (its row is uniform across the wave: the yn read is a broadcast).  g0: an opaque copy of the row group made at
    // the start of every pass -- as loop invariants the RP row coordinates would be hoisted out of the IRLS loop and kept live next to
    // the samples (hundreds of spills)
    auto resid_of = [&](int g0, int u, float zz) -> float {
        const float yn = s_yn[g0 + groups * u];                      // < FIT_YTAB (checked by the launcher)
        const float fit = __fadd_rn(fmaf(yn, Bt, At), __fmul_rn(coef[5], __fmul_rn(yn, yn)));
        return __fsub_rn(zz, fit);
    };
    auto each = [&](auto body) {
        int g0 = grp;
        asm volatile("" : "+s"(g0));
#pragma unroll
        for (int u = 0; u < RP; u++) {
            // the test is on the RESIDUAL (NaN for a sample that is not fitted), not on the sample: "is zr[u] finite" is invariant across
            // the IRLS loop, so the compiler kept all RP answers as exec masks in SGPRs (450 of them spilled to VGPR lanes and read
            // back with two v_readlane per sample and sweep)
            float r = resid_of(g0, u, zr[u]);
            if (finitef(r)) {
                if (mode) r = fabsf(__fsub_rn(r, med));
                body(f2key(r));
            }
        }
    };

    float csig = 1.f;
    for (int it = 0; do_fit && it < iters; it++) {
        // ---- column sums P_b = sum w^2 yn^b (b = 0..4), Q_b = sum w^2 z yn^b (b = 0..2), float64
        double Pb[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, Qb[3] = {0.0, 0.0, 0.0};
        const float inv_csig = __fdiv_rn(1.0f, csig);
        int g0 = grp;
        asm volatile("" : "+s"(g0));
#pragma unroll
        for (int u = 0; u < RP; u++) {
            const float zz = zr[u];
            if (finitef(__fmul_rn(zz, inv_csig))) {               // (finite exactly when zz is; not invariant across the IRLS loop, see `each`)
                float wt = 1.f;
                if (it > 0) {
                    const float uu = __fmul_rn(resid_of(g0, u, zz), inv_csig);
                    wt = __fdiv_rn(1.0f, __fadd_rn(1.0f, __fmul_rn(uu, uu)));
                }
                const double yd = (double)s_yn[g0 + groups * u];
                const double w2 = (double)wt * (double)wt;
                const double t1 = w2 * yd, t2 = t1 * yd, t3 = t2 * yd, t4 = t3 * yd;
                Pb[0] += w2; Pb[1] += t1; Pb[2] += t2; Pb[3] += t3; Pb[4] += t4;
                const double zw = w2 * (double)zz;
                Qb[0] += zw; Qb[1] = fma(zw, yd, Qb[1]); Qb[2] = fma(zw, yd * yd, Qb[2]);
            }
        }
        // the 15 monomial sums m[a][b] = sum x^a P_b (a + b <= 4) and the 6 right-hand sides, reduced over the workgroup
        double v21[21];
        {
            const double x1 = (double)xn, x2 = x1 * x1, x3 = x2 * x1, x4 = x2 * x2;
            const double xp[5] = {1.0, x1, x2, x3, x4};
            int k = 0;
#pragma unroll
            for (int bb = 0; bb <= 4; bb++)
#pragma unroll
                for (int aa = 0; aa + bb <= 4; aa++) v21[k++] = xp[aa] * Pb[bb];     // k = index of (a, b), b-major
            v21[15] = x1 * Qb[0]; v21[16] = Qb[1]; v21[17] = Qb[0]; v21[18] = x2 * Qb[0]; v21[19] = x1 * Qb[1]; v21[20] = Qb[2];
        }
        SEL_STAMP(6);
#pragma unroll
        for (int i = 0; i < 21; i++) {
            const double v = wave_sum(v21[i]);
            if (lane == 0) s_part[wid][i] = v;
        }
        __syncthreads();
        SEL_STAMP(11);
        if (tid < 21) {
            double v = 0.0;
            for (int k = 0; k < NT / 64; k++) v += s_part[k][tid];
            s_sum[tid] = v;
        }
        __syncthreads();
        SEL_STAMP(12);
        if (tid == 0) {
            // monomial (a, b) -> index in v21: b-major with 5, 4, 3, 2, 1 entries
            auto mono = [&](int a, int bb) -> double { const int base[5] = {0, 5, 9, 12, 14}; return s_sum[base[bb] + a]; };
            const int ea[6] = {1, 0, 0, 2, 1, 0}, eb[6] = {0, 1, 0, 0, 1, 2};          // exponents of the basis [x, y, 1, x^2, xy, y^2]
            double A[6][6], rhs[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int j = 0; j < 6; j++) A[i][j] = mono(ea[i] + ea[j], eb[i] + eb[j]);
                rhs[i] = s_sum[15 + i];
            }
            bool ok = nc == 6 ? chol_solve<6>(A, rhs) : chol_solve<3>(A, rhs);
#pragma unroll
            for (int i = 0; i < 6; i++) s_coef[i] = (ok && i < nc) ? (float)rhs[i] : 0.f;
        }
        SEL_STAMP(13);
        __syncthreads();
        for (int i = 0; i < 6; i++) coef[i] = s_coef[i];
        At = fmaf(coef[3], __fmul_rn(xn, xn), fmaf(coef[0], xn, coef[2]));
        Bt = fmaf(coef[4], xn, coef[1]);
        SEL_STAMP(7);
        if (it == iters - 1) break;   // the weights of the last iteration are never used upstream
        // ---- sigma = 1.4826 * (median |r - median r| + 1e-6); the value ranges come from bounds on |fit| (|xn|, |yn| <= 1)
        uint32_t kmin, kmax;
        mode = 0;
        {
            float fb = 0.f;
            for (int i = 0; i < 6; i++) fb += fabsf(coef[i]);
            fb = fb * 1.0001f + 1e-30f;
            kmin = f2key(zmin - fb - 1e-6f * fabsf(zmin)); kmax = f2key(zmax + fb + 1e-6f * fabsf(zmax));
        }
        const float medr = block_median_each<NT>(each, sh, n, kmin, kmax);
        __syncthreads();
        med = medr; mode = 1;
        {
            float hi1 = fabsf(__fsub_rn(key2f(kmax), medr)), hi2 = fabsf(__fsub_rn(key2f(kmin), medr));
            kmin = f2key(0.f); kmax = f2key(hi1 > hi2 ? hi1 : hi2);
        }
        float mad = block_median_each<NT>(each, sh, n, kmin, kmax);
        __syncthreads();
        mode = 0;
        mad = __fadd_rn(mad, 1e-6f);
        csig = __fmul_rn(c, __fmul_rn(1.4826f, mad));
    }
    if (tid < 6) coef_out[b * 6 + tid] = do_fit ? coef[tid] : 0.f;
    // residual plane: z - fit over the WHOLE plane, unfitted pixels included (fit evaluated as eval_poly2d does, :1093-1097, :1132-1135)
    float *out = resid_all + b * (size_t)P;
    if (owner) {
        // fitted pixels still sit in the registers; the others (masked out or NaN upstream) are read again -- all of those loads first, in
        // batches of independent requests (one dependent load per row made this loop a tenth of the kernel), then arithmetic and stores
#pragma unroll
        for (int u = 0; u < RP; u++) {
            const int y = grp + groups * u;
            if (y < h && !finitef(zr[u])) zr[u] = z[(size_t)y * w + col];
        }
#pragma unroll
        for (int u = 0; u < RP; u++) {
            const int y = grp + groups * u;
            if (y < h) {
                float fit = 0.f;
                if (do_fit) {
                    const float yn = s_yn[y];
                    fit = __fadd_rn(__fadd_rn(__fmul_rn(coef[0], xn), __fmul_rn(coef[1], yn)), coef[2]);
                    if (order >= 2) {
                        fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(coef[3], xn), xn));
                        fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(coef[4], xn), yn));
                        fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(coef[5], yn), yn));
                    }
                }
                out[(size_t)y * w + col] = __fsub_rn(zr[u], fit);
            }
        }
    }
    SEL_STAMP(9);
}

#ifdef VISTAF_DEBUG
void fit_debug_dump()
{
    unsigned long long d[16];
    if (hipMemcpyFromSymbol(d, HIP_SYMBOL(g_fit_dbg), sizeof(d)) != hipSuccess) return;
    printf("[fit dbg] frame 0, %llu launches so far, cycle sums: load %llu | column sums %llu | reduce+solve %llu | hist clear %llu | hist pass %llu | bucket search %llu | "
           "collect %llu | sort %llu | second statistic pass %llu | residual plane %llu || of reduce+solve: wave sums %llu | partials %llu | solve %llu | rest in [reduce+solve]\n", d[10], d[8], d[6], d[7], d[0], d[1], d[2], d[3], d[4], d[5], d[9], d[11], d[12], d[13]);
}
#endif

#define VF_FIT_ARGS const float *__restrict__ z_all, const uint8_t *__restrict__ mask_all, int order, int iters, float c, int min_count, int min_mask_count, \
                    float *__restrict__ coef_out, float *__restrict__ resid_all, int h, int w, int cols_pad, int groups
#define VF_FIT_PASS z_all, mask_all, order, iters, c, min_count, min_mask_count, coef_out, resid_all, h, w, cols_pad, groups
template <int RP, int GT>
__global__ __launch_bounds__(SEL_T) void k_robust_polyfit_col(VF_FIT_ARGS) { robust_polyfit_col_body<RP, SEL_T, GT>(VF_FIT_PASS); }
// register-capped variant (5 waves per SIMD = 96 VGPRs, the rest of the samples' working set spills to scratch): its workgroup fits on a CU
// NEXT TO the one-wave march (72 VGPRs, 111 KB of LDS) or flood (64, 109 KB) of another session in flight, where the 128-VGPR workgroup
// needs every register of the CU and has to wait for the march / flood to drain
template <int RP, int GT>
__global__ __launch_bounds__(SEL_T) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_robust_polyfit_col_w5(VF_FIT_ARGS) { robust_polyfit_col_body<RP, SEL_T, GT>(VF_FIT_PASS); }
#undef VF_FIT_ARGS
#undef VF_FIT_PASS

// min_count: 200 fitted pixels upstream (:1103); min_mask_count: 500 mask pixels for the debug_ramp call (shape_ftp.py:1364-1366), else 0
// capped: 1 = prefer the register-capped variant where it exists (RP 56, four row groups), 0 = 128-VGPR variants only
void launch_robust_polyfit(const float *z, const uint8_t *mask, int order, int iters, float c, int min_count, int min_mask_count, float *coef_out,
                           float *resid_out, int B, int h, int w, hipStream_t st, int capped, void *big_scratch)
{
    if (big_scratch && big_frames(B, h * w)) {                   // large frames: every sweep over all pixels of the batch (k_big.hip)
        launch_robust_polyfit_big(z, mask, order, iters, c, min_count, min_mask_count, coef_out, resid_out, B, h, w, big_scratch, st);
        return;
    }
    // i / w == umulhi(i, magic) for every i < h * w as long as h * w * w < 2^32
    const uint32_t magic = ((unsigned long long)h * w * w < 0x100000000ull) ? (uint32_t)(0x100000000ull / (unsigned)w) + 1u : 0u;
    const int cols_pad = ((w + 63) / 64) * 64;
#define VF_FIT_COL(KERNEL, NTV, GR) hipLaunchKernelGGL(KERNEL, dim3(B), dim3(NTV), 0, st, z, mask, order, iters, c, min_count, min_mask_count, coef_out, resid_out, h, w, cols_pad, GR)
    constexpr int NT = SEL_T;                      // threads of every column kernel below (robust_polyfit_col_body's NT): the row groups follow from it
    const int groups = cols_pad <= NT ? std::min(NT / cols_pad, h) : 0;
    const int need = groups ? (h + groups - 1) / groups : 1 << 30;
    if (need <= 64 && h + groups * 16 <= FIT_YTAB) {
        if (capped == 1 && groups == 4 && need > 48 && need <= 56) VF_FIT_COL((k_robust_polyfit_col_w5<56, 4>), SEL_T, 4);
        else if (groups == 4 && need > 32) {
            if (need <= 48) VF_FIT_COL((k_robust_polyfit_col<48, 4>), SEL_T, 4);
            else if (need <= 56) VF_FIT_COL((k_robust_polyfit_col<56, 4>), SEL_T, 4);
            else VF_FIT_COL((k_robust_polyfit_col<64, 4>), SEL_T, 4);
        } else if (need <= 16) VF_FIT_COL((k_robust_polyfit_col<16, 0>), SEL_T, groups);
        else if (need <= 32) VF_FIT_COL((k_robust_polyfit_col<32, 0>), SEL_T, groups);
        else if (need <= 48) VF_FIT_COL((k_robust_polyfit_col<48, 0>), SEL_T, groups);
        else if (need <= 56) VF_FIT_COL((k_robust_polyfit_col<56, 0>), SEL_T, groups);
        else VF_FIT_COL((k_robust_polyfit_col<64, 0>), SEL_T, groups);
        return;
    }
#undef VF_FIT_COL
    hipLaunchKernelGGL(k_robust_polyfit, dim3(B), dim3(SEL_T), 0, st, z, mask, order, iters, c, min_count, min_mask_count, coef_out, resid_out, h, w, magic);
}

}  // namespace vf
