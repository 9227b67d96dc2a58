// Exact order statistics per frame (one 1024-thread workgroup per frame).
//
// Replaces np.percentile / np.nanpercentile / np.median / np.nanmedian on the masked, finite values
// of a plane (shape_ftp.py:343-354, :618-622, :1124-1125).  The two neighbouring order statistics
// are found EXACTLY (range-adaptive 2048-bucket histogram refinement over order-preserving uint32
// keys, then rank-by-counting of <=1024 candidates in LDS); the interpolation then follows NumPy
// 2.x's float32 arithmetic (numpy/lib/_function_base_impl.py `_quantile`, `_lerp`) so thresholds
// equal the CPU path's to the last bit whenever the inputs do.
#pragma once
#include "common.hpp"

namespace vf {

#ifndef SEL_STAMP
#define SEL_STAMP(i) do { } while (0)      // diagnostics hook (k_fit.hip, -DVISTAF_DEBUG)
#endif

constexpr int SEL_T = 1024;
constexpr int SEL_BITS = 13;          // histogram digits: 8192 buckets, 8 per thread in the bucket search
constexpr int SEL_NB = 1 << SEL_BITS;
constexpr int SEL_CAND = 1024;

struct SelShared {
    uint32_t hist[SEL_NB];
    uint32_t cand[SEL_CAND];
    unsigned long long red64[16];
    uint32_t wsum[16];
    uint32_t s_bucket, s_before, s_cnt, s_ncand, s_a, s_b, s_found;
};

// Visit every valid element (get(i, key) -> bool valid), SEL_U elements per thread at a time: the SEL_U loads of a
// batch are issued together, so a pass costs P / (SEL_T * SEL_U) memory round trips per thread instead of P / SEL_T.
constexpr int SEL_U = 8;
template <class F, class B>
__device__ inline void sel_foreach(F get, int P, B body)
{
    int i = threadIdx.x;
    for (; i + (SEL_U - 1) * SEL_T < P; i += SEL_U * SEL_T) {
        uint32_t key[SEL_U];
        bool ok[SEL_U];
#pragma unroll
        for (int u = 0; u < SEL_U; u++) ok[u] = get(i + u * SEL_T, key[u]);
#pragma unroll
        for (int u = 0; u < SEL_U; u++)
            if (ok[u]) body(key[u]);
    }
    for (; i < P; i += SEL_T) {
        uint32_t key;
        if (get(i, key)) body(key);
    }
}

// inclusive prefix sum over the 64 lanes of a wave (DPP row shifts + row broadcasts)
__device__ inline uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// Finds keys of rank k and k+1 (ascending, 0-based) among the n valid elements.
// each(body) calls body(key) for every valid element of this thread (the same elements on every call).
// n must be > 0 and k < n.  Result in all threads.
// NT: threads of the workgroup (a power of two, 256 <= NT <= 1024; the candidate list is limited to NT entries, one per thread of the sort).
template <int NT = SEL_T, class Each>
__device__ __attribute__((always_inline)) inline void block_select2_each(Each each, uint32_t n, uint32_t k, SelShared &sh, uint32_t kmin, uint32_t kmax,
                                                                         uint32_t &key_a, uint32_t &key_b)
{
    static_assert(NT >= 256 && NT <= SEL_T && (NT & (NT - 1)) == 0, "workgroup size");
    constexpr uint32_t CAND = NT < SEL_CAND ? NT : SEL_CAND;
    const int tid = threadIdx.x;
    uint32_t lo = kmin, hi = kmax, below = 0;
    uint32_t a = 0, b = 0;
    for (;;) {
        uint32_t range = hi - lo;
        if (range == 0) { a = lo; break; }
        int bits = 32 - __clz(range);
        int shift = bits > SEL_BITS ? bits - SEL_BITS : 0;
        for (int i = tid; i < SEL_NB; i += NT) sh.hist[i] = 0;
        __syncthreads();
        SEL_STAMP(0);
        each([&](uint32_t key) { if (key >= lo && key <= hi) atomicAdd(&sh.hist[(key - lo) >> shift], 1u); });
        __syncthreads();
        SEL_STAMP(1);
        // locate the bucket holding rank (k - below): each thread owns SEL_NB / SEL_T consecutive buckets
        uint32_t want = k - below;
        constexpr int NBT = SEL_NB / NT;
        uint32_t cb[NBT];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < NBT; j++) { cb[j] = sh.hist[NBT * tid + j]; mine += cb[j]; }
        // exclusive prefix over threads: wave scan + wave totals
        int lane = tid & 63, wid = tid >> 6;
        uint32_t incl = wave_scan_add(mine);
        if (lane == 63) sh.wsum[wid] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int i = 0; i < wid; i++) wbase += sh.wsum[i];
        uint32_t excl = wbase + incl - mine;
        if (want >= excl && want < excl + mine) {
            uint32_t run = excl;
#pragma unroll
            for (int j = 0; j < NBT; j++) {
                if (want >= run && want < run + cb[j]) { sh.s_bucket = NBT * tid + j; sh.s_before = run; sh.s_cnt = cb[j]; }
                run += cb[j];
            }
        }
        __syncthreads();
        uint32_t bsel = sh.s_bucket, cnt = sh.s_cnt;
        below += sh.s_before;
        uint32_t nlo = lo + (bsel << shift);
        uint32_t nhi = shift ? nlo + ((1u << shift) - 1u) : nlo;
        if (nhi > hi || nhi < nlo) nhi = hi;
        lo = nlo; hi = nhi;
        __syncthreads();
        SEL_STAMP(2);
        if (shift == 0) { a = lo; break; }
        if (cnt <= CAND) {
            // collect candidates of this bucket, rank by counting
            if (tid == 0) sh.s_ncand = 0;
            __syncthreads();
            each([&](uint32_t key) {
                // one counter update per wave and visit: the lanes holding a candidate take consecutive slots
                const bool in = key >= lo && key <= hi;
                const unsigned long long mk = __ballot(in);
                if (in) {
                    const int ln = threadIdx.x & 63, leader = __ffsll((long long)mk) - 1;
                    uint32_t base = 0;
                    if (ln == leader) base = atomicAdd(&sh.s_ncand, (uint32_t)__popcll(mk));
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                    const uint32_t pos = base + (uint32_t)__popcll(mk & ((1ull << ln) - 1ull));
                    if (pos < CAND) sh.cand[pos] = key;
                }
            });
            __syncthreads();
            SEL_STAMP(3);
            uint32_t m = sh.s_ncand;
            const uint32_t want2 = k - below;
            if (tid == 0) { sh.s_found = 0; }
            __syncthreads();
            if (m > CAND) m = CAND;                                  // cannot happen (cnt <= SEL_CAND); keeps the indices in range
            // sort the candidates (bitonic network in LDS, padded with the largest key to a power of two >= 64): the two order
            // statistics are then read off by index.  Partners closer than 64 live in the same wave, whose LDS operations execute
            // in order, so only the wider exchanges and the start of a new merge round need a workgroup barrier.
            const uint32_t np2 = m <= 64u ? 64u : 1u << (32 - __clz(m - 1u));
            if ((uint32_t)tid >= m && (uint32_t)tid < np2) sh.cand[tid] = 0xFFFFFFFFu;
            __syncthreads();
            for (uint32_t k2 = 2; k2 <= np2; k2 <<= 1) {
                for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                    const uint32_t ixj = (uint32_t)tid ^ j;
                    if ((uint32_t)tid < np2 && ixj > (uint32_t)tid) {
                        const uint32_t x = sh.cand[tid], y = sh.cand[ixj];
                        const bool up = ((uint32_t)tid & k2) == 0u;
                        if ((x > y) == up) { sh.cand[tid] = y; sh.cand[ixj] = x; }
                    }
                    if (j >= 64u || (j == 1u && k2 >= 64u)) __syncthreads();
                    else __builtin_amdgcn_wave_barrier();
                }
            }
            if (tid == 0) {
                sh.s_a = sh.cand[want2];
                if (want2 + 1 < m) { sh.s_b = sh.cand[want2 + 1]; sh.s_found = 1; }
            }
            __syncthreads();
            SEL_STAMP(4);
            a = sh.s_a;
            if (sh.s_found) { key_a = a; key_b = sh.s_b; __syncthreads(); return; }
            break;
        }
    }
    // b = next order statistic: a again if duplicated, else the smallest key above a
    {
        uint32_t le = 0;
        unsigned long long nxt = ~0ull;
        each([&](uint32_t key) { le += key <= a; if (key > a && (unsigned long long)key < nxt) nxt = key; });
        __syncthreads();
        uint32_t tot = block_sum<uint32_t>(le, sh.wsum);
        unsigned long long mn = block_min_u64(nxt, sh.red64);
        b = (k + 1 < tot || mn == ~0ull) ? a : (uint32_t)mn;
        __syncthreads();
        SEL_STAMP(5);
    }
    key_a = a; key_b = b;
}

// get(i, key) -> bool valid over the P elements of a plane
template <class F>
__device__ inline void block_select2(F get, int P, uint32_t n, uint32_t k, SelShared &sh, uint32_t kmin, uint32_t kmax,
                                     uint32_t &key_a, uint32_t &key_b)
{
    block_select2_each<SEL_T>([&](auto body) { sel_foreach(get, P, body); }, n, k, sh, kmin, kmax, key_a, key_b);
}

// count / min / max of valid keys
template <class F>
__device__ inline void block_minmax(F get, int P, SelShared &sh, uint32_t &n, uint32_t &kmin, uint32_t &kmax)
{
    uint32_t c = 0;
    unsigned long long mn = ~0ull, mx = 0;
    sel_foreach(get, P, [&](uint32_t key) { c++; if (key < mn) mn = key; if (key + 1ull > mx) mx = key + 1ull; });
    __syncthreads();
    n = block_sum<uint32_t>(c, sh.wsum);
    mn = block_min_u64(mn, sh.red64);
    mx = block_max_u64(mx, sh.red64);
    kmin = (uint32_t)mn; kmax = mx ? (uint32_t)(mx - 1) : 0;
    __syncthreads();
}

// NumPy 2.x percentile (method 'linear') of float32 data, float32 arithmetic:
//   vi = n*q + (1 + q*(-1)) - 1 ; prev = floor(vi) ; gamma = vi - prev
//   res = a + (b-a)*gamma  [ b - (b-a)*(1-gamma) when gamma >= 0.5 ]
__device__ inline void np_percentile_index(uint32_t n, float q32, uint32_t &k, float &gamma, bool &top)
{
    float vi = __fsub_rn(__fadd_rn(__fmul_rn((float)n, q32), __fadd_rn(1.0f, __fmul_rn(q32, -1.0f))), 1.0f);
    top = vi >= (float)(n - 1);
    if (vi < 0.0f) { k = 0; gamma = 0.0f; return; }
    float pf = floorf(vi);
    k = (uint32_t)pf;
    gamma = __fsub_rn(vi, pf);
    if (top) { k = n - 1; gamma = 0.0f; }
}
__device__ inline float np_lerp(float a, float b, float t)
{
    float d = __fsub_rn(b, a);
    float r = __fadd_rn(a, __fmul_rn(d, t));
    if (t >= 0.5f) r = __fsub_rn(b, __fmul_rn(d, __fsub_rn(1.0f, t)));
    return r;
}

// percentile (q32 = float32(q)/float32(100)) of valid elements; NaN when n == 0
template <class F>
__device__ inline float block_percentile(F get, int P, float q32, SelShared &sh, uint32_t n, uint32_t kmin, uint32_t kmax)
{
    if (n == 0) return __uint_as_float(0x7fc00000u);
    uint32_t k; float g; bool top;
    np_percentile_index(n, q32, k, g, top);
    uint32_t ka, kb;
    if (k + 1 >= n) {  // last element: both neighbours are the maximum
        if (k >= n) k = n - 1;
        return key2f(kmax);
    }
    block_select2(get, P, n, k, sh, kmin, kmax, ka, kb);
    return np_lerp(key2f(ka), key2f(kb), g);
}

// np.median of valid elements (mean of the two middle values in float32 for even n)
template <int NT = SEL_T, class Each>
__device__ __attribute__((always_inline)) inline float block_median_each(Each each, SelShared &sh, uint32_t n, uint32_t kmin, uint32_t kmax)
{
    if (n == 0) return __uint_as_float(0x7fc00000u);
    if (n == 1) return key2f(kmin);
    uint32_t ka, kb;
    block_select2_each<NT>(each, n, (n & 1u) ? (n - 1) / 2 : n / 2 - 1, sh, kmin, kmax, ka, kb);
    return (n & 1u) ? key2f(ka) : __fdiv_rn(__fadd_rn(key2f(ka), key2f(kb)), 2.0f);
}
template <class F>
__device__ inline float block_median(F get, int P, SelShared &sh, uint32_t n, uint32_t kmin, uint32_t kmax)
{
    return block_median_each<SEL_T>([&](auto body) { sel_foreach(get, P, body); }, sh, n, kmin, kmax);
}

}  // namespace vf
