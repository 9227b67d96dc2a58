// Exact order statistics and the IRLS polynomial fit for LARGE frames (the native 1182 x 1182 crops), as chains of streaming kernels.
//
// k_select / k_robust_polyfit keep one 1024-thread workgroup per frame busy with ~3 / ~100 sweeps over the frame: at 224 x 224 x 256 that is
// one frame per CU and fills the chip; at native size a batch of 8 frames leaves 248 of the 256 CUs idle (detrend 46 ms, eight selections
// ~1.5 ms each).  Here every sweep is a kernel over ALL pixels of the batch and the decisions between sweeps are single-workgroup kernels on
// per-frame state in global memory -- no host round trip, no grid barrier:
//   selection (np.percentile / np.median semantics of select.hpp, same keys, same interpolation):
//     range (one sweep, or bounds handed in) -> per level: 2048-bucket histogram of the keys in [lo, hi] (LDS histograms per block, merged
//     with atomics; integer counts, so the result does not depend on the order) -> bucket of the wanted rank -> narrower [lo, hi]; the
//     bucket width shrinks 32 -> 21 -> 10 -> 0 bits, so three levels always end at ONE key; the next order statistic comes out of that last
//     level too (its histogram has one bucket per key; its sweep also keeps the smallest key above the range).
//   robust_polyfit2d (shape_ftp.py:1100-1136): per IRLS step the 21 normal-equation sums as per-block partials reduced in a FIXED order
//     (deterministic), float64 Cholesky on one thread, then the two medians through the selection above with the residual as the value.
// Arithmetic per sample, keys, median / percentile interpolation and the residual plane are those of k_fit.hip / select.hpp.
#include <algorithm>
#include <cstdio>
#include "kernels.hpp"
#include "select.hpp"

namespace vf {

namespace {

constexpr int SB_BITS = 11, SB_NB = 1 << SB_BITS;          // buckets per level
constexpr int SB_MAXREQ = 4;
constexpr int SB_T = 256;                                   // threads of the sweep kernels
constexpr int SB_PX = 16;                                   // pixels per thread and sweep block

struct SbReq {                      // one (frame, request)
    uint32_t lo, hi, below, k, shift, done, a, b, cnt_le, min_gt, two, pad;
    float q, gamma, result, padf;
};
struct SbFrame { uint32_t n, kmin, kmax, nmask; };

// ---- value sources -------------------------------------------------------------------------------------------------------------------
struct PlaneSrc {                   // k_select's PlaneGetter
    static constexpr bool tiled = false;
    const float *v; const uint8_t *m; size_t mstride; const float *le; int use_abs; int P;
    __device__ bool key(size_t b, int i, uint32_t &k) const
    {
        const uint8_t mk = m[b * mstride + i];
        float x = v[b * (size_t)P + i];
        bool ok = mk != 0 && finitef(x);
        if (use_abs) x = fabsf(x);
        if (le && !(x <= le[b])) ok = false;
        k = f2key(x);
        return ok;
    }
};
struct FitState {                   // per frame
    float coef[6];
    float med, csig, zmin, zmax;
    int do_fit, mode, n, pad;
};
// residual (mode 0) or |residual - med| (mode 1) of the fitted samples, k_fit.hip's arithmetic.  Tiled sweeps (sb_visit): a thread keeps one
// image column for SB_PX rows, so the terms that only depend on x -- the normalised coordinate, A(x) = c3 x^2 + c0 x + c2 and
// B(x) = c4 x + c1 -- are formed once per block instead of once per pixel (with the pixel's row by an integer division on top): a sample is
// then fit = A + y B + c5 y^2 exactly as in the column kernels of k_fit.hip.
struct ResidSrc {
    static constexpr bool tiled = true;
    const float *z; const uint8_t *m; const FitState *fs; int h, w, tiles_x;
    struct Col { float At, Bt, c5, med, cyf; int mode; };
    __device__ Col column(size_t b, int x) const
    {
        const FitState &s = fs[b];
        const float cxf = (float)((w - 1) / 2.0);
        const float xn = __fdiv_rn(__fsub_rn((float)x, cxf), cxf);
        Col c;
        c.At = fmaf(s.coef[3], __fmul_rn(xn, xn), fmaf(s.coef[0], xn, s.coef[2]));
        c.Bt = fmaf(s.coef[4], xn, s.coef[1]);
        c.c5 = s.coef[5]; c.med = s.med; c.mode = s.mode; c.cyf = (float)((h - 1) / 2.0);
        return c;
    }
    __device__ bool key(const Col &c, size_t b, int y, int x, uint32_t &k) const
    {
        const size_t i = b * (size_t)h * w + (size_t)y * w + x;
        const float zz = z[i];
        if (!m[i] || !finitef(zz)) return false;
        const float yn = __fdiv_rn(__fsub_rn((float)y, c.cyf), c.cyf);
        const float fit = __fadd_rn(fmaf(yn, c.Bt, c.At), __fmul_rn(c.c5, __fmul_rn(yn, yn)));
        float r = __fsub_rn(zz, fit);
        if (c.mode) r = fabsf(__fsub_rn(r, c.med));
        k = f2key(r);
        return true;
    }
};

// body(key) for every valid pixel of this block's share of frame b: SB_T * SB_PX consecutive pixels, or a tile of SB_T columns x SB_PX rows
template <class Src, class Body>
__device__ __attribute__((always_inline)) inline void sb_visit(const Src &src, size_t b, int P, Body body)
{
    if constexpr (Src::tiled) {
        const int ty = blockIdx.x / src.tiles_x, tx = blockIdx.x - ty * src.tiles_x;
        const int x = tx * SB_T + threadIdx.x, y0 = ty * SB_PX;
        const bool col_ok = x < src.w;
        const typename Src::Col c = src.column(b, col_ok ? x : 0);
#pragma unroll 4
        for (int u = 0; u < SB_PX; u++) {
            const int y = y0 + u;
            uint32_t k;
            if (col_ok && y < src.h && src.key(c, b, y, x, k)) body(k);
        }
    } else {
        const int i0 = blockIdx.x * SB_T * SB_PX + threadIdx.x;
#pragma unroll 4
        for (int u = 0; u < SB_PX; u++) {
            const int i = i0 + u * SB_T;
            uint32_t k;
            if (i < P && src.key(b, i, k)) body(k);
        }
    }
}

// ---- selection kernels ---------------------------------------------------------------------------------------------------------------
__global__ void k_sb_frame_init(SbFrame *fr, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { fr[b].n = 0; fr[b].kmin = 0xFFFFFFFFu; fr[b].kmax = 0; fr[b].nmask = 0; }
}
template <class Src>
__global__ __launch_bounds__(SB_T) void k_sb_minmax(Src src, SbFrame *fr, int P)
{
    __shared__ unsigned long long s64[16];
    __shared__ uint32_t s32[16];
    const size_t b = blockIdx.y;
    uint32_t c = 0;
    unsigned long long mn = ~0ull, mx = 0;
    sb_visit(src, b, P, [&](uint32_t k) { c++; if (k < mn) mn = k; if (k + 1ull > mx) mx = k + 1ull; });
    __syncthreads();
    const uint32_t n = block_sum<uint32_t>(c, s32);
    mn = block_min_u64(mn, s64);
    mx = block_max_u64(mx, s64);
    if (threadIdx.x == 0 && n) { atomicAdd(&fr[b].n, n); atomicMin(&fr[b].kmin, (uint32_t)mn); atomicMax(&fr[b].kmax, (uint32_t)(mx - 1)); }
}

// first level of every request: rank, interpolation weight, bucket width; median: q < 0
__global__ void k_sb_setup(SbReq *rq, const SbFrame *fr, const float *reqs, int nreq, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * nreq) return;
    const int b = t / nreq, j = t - b * nreq;
    SbReq r = {};
    r.min_gt = 0xFFFFFFFFu;
    const SbFrame f = fr[b];
    r.q = reqs ? reqs[j] : -1.f;
    r.lo = f.kmin; r.hi = f.kmax;
    const uint32_t n = f.n;
    if (n == 0) { r.done = 2; r.result = __uint_as_float(0x7fc00000u); }
    else if (r.q < 0.f) {                                         // np.median
        if (n == 1) { r.done = 2; r.result = key2f(f.kmin); }
        else { r.k = (n & 1u) ? (n - 1) / 2 : n / 2 - 1; r.two = (n & 1u) ? 0u : 1u; r.gamma = 0.5f; }
    } else {                                                      // np.percentile, method linear (block_percentile)
        uint32_t k; float g; bool top;
        np_percentile_index(n, r.q, k, g, top);
        if (k + 1 >= n) { r.done = 2; r.result = key2f(f.kmax); }
        else { r.k = k; r.gamma = g; r.two = 2u; }
    }
    if (!r.done) {
        const uint32_t range = r.hi - r.lo;
        const int bits = range ? 32 - __clz(range) : 0;
        r.shift = bits > SB_BITS ? (uint32_t)(bits - SB_BITS) : 0u;
    }
    rq[t] = r;
}

// At a level whose buckets are one key wide (the last one) the sweep also keeps the smallest key ABOVE the range: the next order statistic
// when the wanted rank is the last key of the range.
template <class Src>
__global__ __launch_bounds__(SB_T) void k_sb_hist(Src src, SbReq *rq, uint32_t *hist, int nreq, int P)
{
    __shared__ uint32_t lh[SB_MAXREQ][SB_NB];
    __shared__ SbReq sr[SB_MAXREQ];
    __shared__ unsigned long long s64[16];
    const size_t b = blockIdx.y;
    if (threadIdx.x < nreq) sr[threadIdx.x] = rq[b * nreq + threadIdx.x];
    for (int i = threadIdx.x; i < nreq * SB_NB; i += SB_T) (&lh[0][0])[i] = 0;
    __syncthreads();
    bool any = false;
    for (int j = 0; j < nreq; j++) any = any || !sr[j].done;
    if (!any) return;
    unsigned long long nx[SB_MAXREQ];
#pragma unroll
    for (int j = 0; j < SB_MAXREQ; j++) nx[j] = ~0ull;
    sb_visit(src, b, P, [&](uint32_t k) {
#pragma unroll
        for (int j = 0; j < SB_MAXREQ; j++)
            if (j < nreq && !sr[j].done) {
                if (k >= sr[j].lo && k <= sr[j].hi) atomicAdd(&lh[j][(k - sr[j].lo) >> sr[j].shift], 1u);
                else if (k > sr[j].hi && k < nx[j]) nx[j] = k;
            }
    });
    __syncthreads();
    for (int i = threadIdx.x; i < nreq * SB_NB; i += SB_T) {
        const uint32_t v = (&lh[0][0])[i];
        if (v) atomicAdd(&hist[(b * nreq) * SB_NB + i], v);
    }
#pragma unroll
    for (int j = 0; j < SB_MAXREQ; j++) {
        if (j >= nreq || sr[j].done || sr[j].shift != 0) continue;          // (uniform)
        __syncthreads();
        const unsigned long long mn = block_min_u64(nx[j], s64);
        if (threadIdx.x == 0 && mn != ~0ull) atomicMin(&rq[b * nreq + j].min_gt, (uint32_t)mn);
    }
}

// bucket of the wanted rank -> next level; once the buckets are one key wide: the key itself, the number of keys <= it and the next key present
// (the first non-empty bucket above, else the smallest key above the range from the sweep).  Clears the histogram again.
__global__ __launch_bounds__(1024) void k_sb_pick(SbReq *rq, uint32_t *hist, int nreq)
{
    __shared__ uint32_t wsum[16], wmin[16];
    __shared__ uint32_t s_bucket, s_before, s_cnt;
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int j = 0; j < nreq; j++) {
        SbReq &r = rq[b * nreq + j];
        uint32_t *hh = hist + (b * nreq + j) * SB_NB;
        if (r.done) continue;
        const uint32_t shift = r.shift;
        const uint32_t c0 = hh[2 * tid], c1 = hh[2 * tid + 1];
        hh[2 * tid] = 0; hh[2 * tid + 1] = 0;
        const uint32_t mine = c0 + c1;
        const uint32_t incl = wave_scan_add(mine);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int i = 0; i < wid; i++) base += wsum[i];
        const uint32_t excl = base + incl - mine, want = r.k - r.below;
        if (want >= excl && want < excl + mine) {
            if (want < excl + c0) { s_bucket = 2 * tid; s_before = excl; s_cnt = c0; }
            else { s_bucket = 2 * tid + 1; s_before = excl + c0; s_cnt = c1; }
        }
        __syncthreads();
        uint32_t nb = 0xFFFFFFFFu;                               // first non-empty bucket above the chosen one
        if (shift == 0) {
            const uint32_t sb = s_bucket;
            if (2u * tid > sb && c0) nb = 2u * tid;
            else if (2u * tid + 1u > sb && c1) nb = 2u * tid + 1u;
            for (int o = 32; o; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)nb, o, 64); nb = v < nb ? v : nb; }
            if (lane == 0) wmin[wid] = nb;
            __syncthreads();
        }
        if (tid == 0) {
            const uint32_t nlo = r.lo + (s_bucket << shift);
            uint32_t nhi = shift ? nlo + ((1u << shift) - 1u) : nlo;
            if (nhi > r.hi || nhi < nlo) nhi = r.hi;
            const uint32_t lo_old = r.lo;
            r.below += s_before; r.lo = nlo; r.hi = nhi;
            if (shift == 0) {
                uint32_t nbm = 0xFFFFFFFFu;
                for (int i = 0; i < 16; i++) nbm = wmin[i] < nbm ? wmin[i] : nbm;
                r.a = nlo; r.done = 1;
                r.cnt_le = r.below + s_cnt;                      // keys <= a
                if (nbm != 0xFFFFFFFFu) r.min_gt = lo_old + nbm; // else: the smallest key above the range (k_sb_hist), 0xFFFFFFFF if none
            } else r.shift = shift > (uint32_t)SB_BITS ? shift - SB_BITS : 0u;
        }
        __syncthreads();
    }
}

__global__ void k_sb_finish(SbReq *rq, float *out, int *counts, const SbFrame *fr, int nreq, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * nreq) return;
    SbReq &r = rq[t];
    if (r.done == 1) {
        uint32_t kb = r.a;
        if (r.two) kb = (r.k + 1 < r.cnt_le || r.min_gt == 0xFFFFFFFFu) ? r.a : r.min_gt;
        const float a = key2f(r.a), bb = key2f(kb);
        if (r.two == 2u) r.result = np_lerp(a, bb, r.gamma);
        else r.result = r.two ? __fdiv_rn(__fadd_rn(a, bb), 2.0f) : a;
        r.b = kb;
    }
    if (out) out[t] = r.result;
    if (counts && t % nreq == 0) counts[t / nreq] = (int)fr[t / nreq].n;
}

// ---- fit kernels ---------------------------------------------------------------------------------------------------------------------
constexpr int FB_T = 256, FB_PX = 16;

// fitted samples / masked pixels / range of z
__global__ __launch_bounds__(FB_T) void k_fb_count(const float *__restrict__ z_all, const uint8_t *__restrict__ m_all, SbFrame *fr, int P)
{
    __shared__ unsigned long long s64[16];
    __shared__ uint32_t s32[16];
    const size_t b = blockIdx.y;
    uint32_t c = 0, cm = 0;
    unsigned long long mn = ~0ull, mx = 0;
    const int i0 = blockIdx.x * FB_T * FB_PX + threadIdx.x;
#pragma unroll
    for (int u = 0; u < FB_PX; u++) {
        const int i = i0 + u * FB_T;
        if (i >= P) continue;
        const uint8_t mk = m_all[b * (size_t)P + i];
        const float zz = z_all[b * (size_t)P + i];
        cm += mk != 0;
        if (mk && finitef(zz)) { const uint32_t k = f2key(zz); c++; if (k < mn) mn = k; if (k + 1ull > mx) mx = k + 1ull; }
    }
    __syncthreads();
    const uint32_t n = block_sum<uint32_t>(c, s32);
    const uint32_t nm = block_sum<uint32_t>(cm, s32);
    mn = block_min_u64(mn, s64);
    mx = block_max_u64(mx, s64);
    if (threadIdx.x == 0) {
        if (nm) atomicAdd(&fr[b].nmask, nm);
        if (n) { atomicAdd(&fr[b].n, n); atomicMin(&fr[b].kmin, (uint32_t)mn); atomicMax(&fr[b].kmax, (uint32_t)(mx - 1)); }
    }
}
__global__ void k_fb_init(FitState *fs, const SbFrame *fr, int min_count, int min_mask_count, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    FitState s = {};
    s.n = (int)fr[b].n;
    s.do_fit = s.n >= min_count && (min_mask_count <= 0 || (int)fr[b].nmask >= min_mask_count);
    s.zmin = s.n ? key2f(fr[b].kmin) : 0.f; s.zmax = s.n ? key2f(fr[b].kmax) : 0.f;
    s.csig = 1.f;
    fs[b] = s;
}
// 21 normal-equation sums of the block's samples: weights w = 1 / (1 + (r / csig)^2) (first step: 1), float64 sums of exact products.
// Blocks are tiles of FB_T columns x FB_PX rows (a thread keeps its column: see ResidSrc).
__global__ __launch_bounds__(FB_T) void k_fb_sums(const float *__restrict__ z_all, const uint8_t *__restrict__ m_all, const FitState *fs, double *partial,
                                                  int it, int h, int w, int tiles_x)
{
    __shared__ double s_part[FB_T / 64][21];
    const size_t b = blockIdx.y;
    const FitState s = fs[b];
    if (!s.do_fit) return;
    const size_t P = (size_t)h * w;
    const float cxf = (float)((w - 1) / 2.0), cyf = (float)((h - 1) / 2.0);
    const float inv_csig = __fdiv_rn(1.0f, s.csig);
    double v[21];
#pragma unroll
    for (int i = 0; i < 21; i++) v[i] = 0.0;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x = tx * FB_T + threadIdx.x, y0 = ty * FB_PX;
    if (x < w) {
        const float xn = __fdiv_rn(__fsub_rn((float)x, cxf), cxf);
        const float At = fmaf(s.coef[3], __fmul_rn(xn, xn), fmaf(s.coef[0], xn, s.coef[2])), Bt = fmaf(s.coef[4], xn, s.coef[1]);
        const double xd = xn;
        for (int u = 0; u < FB_PX; u++) {
            const int y = y0 + u;
            if (y >= h) break;
            const size_t i = b * P + (size_t)y * w + x;
            const float zz = z_all[i];
            if (!m_all[i] || !finitef(zz)) continue;
            const float yn = __fdiv_rn(__fsub_rn((float)y, cyf), cyf);
            float wt = 1.f;
            if (it > 0) {
                const float fit = __fadd_rn(fmaf(yn, Bt, At), __fmul_rn(s.coef[5], __fmul_rn(yn, yn)));
                const float uu = __fmul_rn(__fsub_rn(zz, fit), inv_csig);
                wt = __fdiv_rn(1.0f, __fadd_rn(1.0f, __fmul_rn(uu, uu)));
            }
            const double w2 = (double)wt * (double)wt, yd = yn, zw = w2 * (double)zz;
            // monomials x^a y^b, b-major with 5, 4, 3, 2, 1 entries (k_fit.hip's index), then the six right-hand sides
            double yp = w2;
            int k = 0;
#pragma unroll
            for (int bb = 0; bb <= 4; bb++) {
                double xp = yp;
#pragma unroll
                for (int aa = 0; aa + bb <= 4; aa++) { v[k++] += xp; xp *= xd; }
                yp *= yd;
            }
            v[15] += zw * xd; v[16] += zw * yd; v[17] += zw; v[18] += zw * xd * xd; v[19] += zw * xd * yd; v[20] += zw * yd * yd;
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 21; i++) { const double sv = wave_sum(v[i]); if (lane == 0) s_part[wid][i] = sv; }
    __syncthreads();
    if (threadIdx.x < 21) {
        double t = 0.0;
        for (int k = 0; k < FB_T / 64; k++) t += s_part[k][threadIdx.x];
        partial[(b * gridDim.x + blockIdx.x) * 21 + threadIdx.x] = t;
    }
}

template <int N>
__device__ inline bool fb_chol(double (&A)[6][6], double (&rhs)[6])
{
    double L[N][N], inv[N];
    for (int i = 0; i < N; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) { if (!(s > 0.0)) return false; L[i][i] = sqrt(s); inv[i] = 1.0 / L[i][i]; }
            else L[i][j] = s * inv[j];
        }
    for (int i = 0; i < N; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i][k] * rhs[k]; rhs[i] = s * inv[i]; }
    for (int i = N - 1; i >= 0; i--) { double s = rhs[i]; for (int k = i + 1; k < N; k++) s -= L[k][i] * rhs[k]; rhs[i] = s * inv[i]; }
    return true;
}
constexpr int FS_CHUNKS = 12;           // 21 * 12 = 252 threads
// partials in block order -> coefficients; the value range of the residuals for the two selections (bounds on |fit|, as k_fit.hip)
__global__ __launch_bounds__(256) void k_fb_solve(FitState *fs, const double *partial, SbFrame *sel_fr, int nblk, int order)
{
    __shared__ double s_sum[21], s_chunk[FS_CHUNKS][21];
    const size_t b = blockIdx.x;
    FitState &s = fs[b];
    if (!s.do_fit) return;
    // every sum in a fixed order: FS_CHUNKS contiguous runs of blocks added up side by side, then the runs in order (one thread walking all
    // ~370 partials of a native crop was 90 us per call)
    if (threadIdx.x < 21 * FS_CHUNKS) {
        const int e = threadIdx.x % 21, c = threadIdx.x / 21;
        const int per = (nblk + FS_CHUNKS - 1) / FS_CHUNKS, k0 = c * per, k1 = min(nblk, k0 + per);
        double t = 0.0;
        for (int k = k0; k < k1; k++) t += partial[(b * nblk + k) * 21 + e];
        s_chunk[c][e] = t;
    }
    __syncthreads();
    if (threadIdx.x < 21) {
        double t = 0.0;
        for (int c = 0; c < FS_CHUNKS; c++) t += s_chunk[c][threadIdx.x];
        s_sum[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nc = order >= 2 ? 6 : 3;
        auto mono = [&](int a, int bb) -> double { const int base[5] = {0, 5, 9, 12, 14}; return s_sum[base[bb] + a]; };
        const int ea[6] = {1, 0, 0, 2, 1, 0}, eb[6] = {0, 1, 0, 0, 1, 2};
        double A[6][6], rhs[6];
        for (int i = 0; i < 6; i++) {
            for (int j = 0; j < 6; j++) A[i][j] = mono(ea[i] + ea[j], eb[i] + eb[j]);
            rhs[i] = s_sum[15 + i];
        }
        const bool ok = nc == 6 ? fb_chol<6>(A, rhs) : fb_chol<3>(A, rhs);
        float fb = 0.f;
        for (int i = 0; i < 6; i++) { s.coef[i] = (ok && i < nc) ? (float)rhs[i] : 0.f; fb += fabsf(s.coef[i]); }
        fb = fb * 1.0001f + 1e-30f;
        s.mode = 0;
        sel_fr[b].n = (uint32_t)s.n;
        sel_fr[b].kmin = f2key(s.zmin - fb - 1e-6f * fabsf(s.zmin));
        sel_fr[b].kmax = f2key(s.zmax + fb + 1e-6f * fabsf(s.zmax));
    }
}
// after the median of r: switch to |r - med| and its range; after the median of |r - med|: the next step's scale
__global__ void k_fb_after_median(FitState *fs, const SbReq *rq, SbFrame *sel_fr, int B, int which, float c)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    FitState &s = fs[b];
    if (!s.do_fit) return;
    if (which == 0) {
        const float medr = rq[b].result;
        const float hi1 = fabsf(__fsub_rn(key2f(sel_fr[b].kmax), medr)), hi2 = fabsf(__fsub_rn(key2f(sel_fr[b].kmin), medr));
        s.med = medr; s.mode = 1;
        sel_fr[b].kmin = f2key(0.f); sel_fr[b].kmax = f2key(hi1 > hi2 ? hi1 : hi2);
    } else {
        const float mad = __fadd_rn(rq[b].result, 1e-6f);
        s.csig = __fmul_rn(c, __fmul_rn(1.4826f, mad));
        s.mode = 0;
    }
}
__global__ __launch_bounds__(256) void k_fb_resid(const float *__restrict__ z_all, const FitState *fs, float *__restrict__ coef_out, float *__restrict__ out_all,
                                                  int order, int h, int w)
{
    const size_t b = blockIdx.y;
    const FitState s = fs[b];
    const int P = h * w;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 6) coef_out[b * 6 + threadIdx.x] = s.do_fit ? s.coef[threadIdx.x] : 0.f;
    if (i >= P) return;
    const float cxf = (float)((w - 1) / 2.0), cyf = (float)((h - 1) / 2.0);
    const int y = i / w, x = i - y * w;
    float fit = 0.f;
    if (s.do_fit) {                                     // eval_poly2d's operation order (:1093-1097)
        const float xn = __fdiv_rn(__fsub_rn((float)x, cxf), cxf), yn = __fdiv_rn(__fsub_rn((float)y, cyf), cyf);
        fit = __fadd_rn(__fadd_rn(__fmul_rn(s.coef[0], xn), __fmul_rn(s.coef[1], yn)), s.coef[2]);
        if (order >= 2) {
            fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(s.coef[3], xn), xn));
            fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(s.coef[4], xn), yn));
            fit = __fadd_rn(fit, __fmul_rn(__fmul_rn(s.coef[5], yn), yn));
        }
    }
    out_all[b * (size_t)P + i] = __fsub_rn(z_all[b * (size_t)P + i], fit);
}

// scratch layout: [B] SbFrame | [B * SB_MAXREQ] SbReq | [B * SB_MAXREQ * SB_NB] u32 histograms | [B] FitState | [B * nblk * 21] double partials
struct BigScratch { SbFrame *fr; SbReq *rq; uint32_t *hist; FitState *fs; double *partial; };
static BigScratch big_carve(void *scratch, int B, int P)
{
    BigScratch s;
    uint8_t *p = (uint8_t *)scratch;
    s.fr = (SbFrame *)p; p += (((size_t)B * sizeof(SbFrame)) + 255) & ~(size_t)255;
    s.rq = (SbReq *)p; p += (((size_t)B * SB_MAXREQ * sizeof(SbReq)) + 255) & ~(size_t)255;
    s.hist = (uint32_t *)p; p += (size_t)B * SB_MAXREQ * SB_NB * 4;
    s.fs = (FitState *)p; p += (((size_t)B * sizeof(FitState)) + 255) & ~(size_t)255;
    s.partial = (double *)p;
    (void)P;
    return s;
}
static int sweep_blocks(int P) { return (P + SB_T * SB_PX - 1) / (SB_T * SB_PX); }
static int tile_blocks(int h, int w) { return ((w + SB_T - 1) / SB_T) * ((h + SB_PX - 1) / SB_PX); }
template <class Src>
static int sweep_grid(const Src &src, int P)
{
    if constexpr (Src::tiled) return tile_blocks(src.h, src.w);
    else return sweep_blocks(P);
}

// the selection proper on a prepared SbFrame (n, kmin, kmax per frame)
template <class Src>
static void select_levels(const Src &src, const BigScratch &S, const float *reqs_dev, int nreq, float *out, int *counts, int B, int P, hipStream_t st)
{
    const dim3 sweep(sweep_grid(src, P), B);
    hipLaunchKernelGGL(k_sb_setup, dim3((B * nreq + 63) / 64), dim3(64), 0, st, S.rq, S.fr, reqs_dev, nreq, B);
    for (int level = 0; level < 3; level++) {
        hipLaunchKernelGGL((k_sb_hist<Src>), sweep, dim3(SB_T), 0, st, src, S.rq, S.hist, nreq, P);
        hipLaunchKernelGGL(k_sb_pick, dim3(B), dim3(1024), 0, st, S.rq, S.hist, nreq);
    }
    hipLaunchKernelGGL(k_sb_finish, dim3((B * nreq + 63) / 64), dim3(64), 0, st, S.rq, out, counts, S.fr, nreq, B);
}

}  // namespace

size_t big_scratch_bytes(int B, int h, int w)
{
    const size_t P = (size_t)h * w;
    return ((((size_t)B * sizeof(SbFrame)) + 255) & ~(size_t)255) + ((((size_t)B * SB_MAXREQ * sizeof(SbReq)) + 255) & ~(size_t)255) +
           (size_t)B * SB_MAXREQ * SB_NB * 4 + ((((size_t)B * sizeof(FitState)) + 255) & ~(size_t)255) +
           (size_t)B * std::max((P + FB_T * FB_PX - 1) / (FB_T * FB_PX), (size_t)tile_blocks(h, w)) * 21 * sizeof(double) + 1024;
}
// frames large enough that a batch cannot fill the chip with one workgroup per frame
bool big_frames(int B, int P) { return P >= 262144 && B <= 192; }

void launch_select_big(const float *vals, const uint8_t *mask, size_t mask_stride, const float *le_thr, bool use_abs, const float *reqs_dev, int nreq,
                       float *out, int *counts, int B, int P, void *scratch, hipStream_t st)
{
    const BigScratch S = big_carve(scratch, B, P);
    PlaneSrc src{vals, mask, mask_stride, le_thr, use_abs ? 1 : 0, P};
    (void)hipMemsetAsync(S.hist, 0, (size_t)B * SB_MAXREQ * SB_NB * 4, st);
    hipLaunchKernelGGL(k_sb_frame_init, dim3((B + 63) / 64), dim3(64), 0, st, S.fr, B);
    hipLaunchKernelGGL(k_sb_minmax<PlaneSrc>, dim3(sweep_blocks(P), B), dim3(SB_T), 0, st, src, S.fr, P);
    select_levels(src, S, reqs_dev, nreq, out, counts, B, P, st);
}

void launch_robust_polyfit_big(const float *z, const uint8_t *mask, int order, int iters, float c, int min_count, int min_mask_count, float *coef_out,
                               float *resid_out, int B, int h, int w, void *scratch, hipStream_t st)
{
    const int P = h * w;
    const BigScratch S = big_carve(scratch, B, P);
    const int nblk = (P + FB_T * FB_PX - 1) / (FB_T * FB_PX);
    (void)hipMemsetAsync(S.hist, 0, (size_t)B * SB_MAXREQ * SB_NB * 4, st);
    hipLaunchKernelGGL(k_sb_frame_init, dim3((B + 63) / 64), dim3(64), 0, st, S.fr, B);
    hipLaunchKernelGGL(k_fb_count, dim3(nblk, B), dim3(FB_T), 0, st, z, mask, S.fr, P);
    hipLaunchKernelGGL(k_fb_init, dim3((B + 63) / 64), dim3(64), 0, st, S.fs, S.fr, min_count, min_mask_count, B);
    static_assert(FB_T == SB_T && FB_PX == SB_PX, "one tile shape");
    const int tiles_x = (w + SB_T - 1) / SB_T, ntile = tile_blocks(h, w);
    ResidSrc src{z, mask, S.fs, h, w, tiles_x};
    for (int it = 0; it < iters; it++) {
        hipLaunchKernelGGL(k_fb_sums, dim3(ntile, B), dim3(FB_T), 0, st, z, mask, S.fs, S.partial, it, h, w, tiles_x);
        hipLaunchKernelGGL(k_fb_solve, dim3(B), dim3(256), 0, st, S.fs, S.partial, S.fr, ntile, order);
        if (it == iters - 1) break;                     // the weights of the last step are never used upstream
        for (int which = 0; which < 2; which++) {
            select_levels(src, S, nullptr, 1, nullptr, nullptr, B, P, st);
            hipLaunchKernelGGL(k_fb_after_median, dim3((B + 63) / 64), dim3(64), 0, st, S.fs, S.rq, S.fr, B, which, c);
        }
    }
    hipLaunchKernelGGL(k_fb_resid, dim3((P + 255) / 256, B), dim3(256), 0, st, z, S.fs, coef_out, resid_out, order, h, w);
}

}  // namespace vf
