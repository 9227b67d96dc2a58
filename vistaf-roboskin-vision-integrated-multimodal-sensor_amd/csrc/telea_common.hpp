// Shared pieces of the LDS-resident Telea march (k_inpaint_win.hip: frame-window and cluster kernels): flag byte layout, DPP helpers, the
// two-run stable priority queue, the FMM quadrant solve and the raster seed scan.
#pragma once
#include "kernels.hpp"

namespace vf {

// flag byte: bits 0-1 state, bit 4 outside the image, bit 5 scratch (hole in row range), bit 6 hole pixel,
// bit 7 seed (initial band)
constexpr uint8_t W_KNOWN = 0, W_BAND = 1, W_INSIDE = 2, W_CHANGE = 3, W_ST = 3, W_BORDER = 0x10, W_ROW = 0x20, W_HOLE = 0x40, W_SEED = 0x80;

__device__ inline float wn_dpp_sum(float x)
{
    int v = __float_as_int(x);
#define VF_ADD(ctrl, rm)                                                                          \
    v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(0, v, ctrl, rm, 0xf, false)));
    VF_ADD(0xB1, 0xf) VF_ADD(0x4E, 0xf) VF_ADD(0x141, 0xf) VF_ADD(0x140, 0xf) VF_ADD(0x142, 0xa) VF_ADD(0x143, 0xc)
#undef VF_ADD
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
__device__ inline float wn_lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// Sum over lanes [0, N) in LANE ORDER from `init`: ((init + x0) + x1) + ... + x(N-1), every addition rounded to float.  That is the order
// in which cv::inpaint's k / l loops accumulate Ia, Jx, Jy and s (inpaint.cpp icvTeleaInpaintFMM), so the
// filled pixels carry the same roundings bit for bit -- they feed the illumination blur, the demodulation and, through the amplitude, the
// quality >= p25 threshold of the reliable mask.  After step t lanes 0..t hold their final prefix (lane l takes lane l-1's value through
// wave_shr:1 and adds its own term; a lane that is already final recomputes the same value), so N-1 dependent DPP adds give the total.
// A tree reduction is ~6 steps instead of N-1 but rounds differently (up to 1e-6 relative on a filled pixel).
template <int N>
__device__ __attribute__((always_inline)) inline float wn_seq_sum(float x, float init, int lane)
{
    const float xi = lane == 0 ? __fadd_rn(init, x) : x;
    float S = xi;
#pragma unroll
    for (int t = 1; t < N; t++)
        S = __fadd_rn(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(S), 0x138, 0xf, 0xf, true)), xi);     // lane 0: 0 + xi
    return wn_lane_f(S, N - 1);
}
// the four estimator sums side by side: four independent chains, so a step issues back to back instead of waiting out the DPP hazard
template <int N>
__device__ __attribute__((always_inline)) inline void wn_seq_sum4(float &a, float &b, float &c, float &d, float init_d, int lane)
{
    const float xd = lane == 0 ? __fadd_rn(init_d, d) : d;      // a, b, c start at +0: 0 + x = x
    float Sa = a, Sb = b, Sc = c, Sd = xd;
#pragma unroll
    for (int t = 1; t < N; t++) {
        const float pa = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Sa), 0x138, 0xf, 0xf, true));
        const float pb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Sb), 0x138, 0xf, 0xf, true));
        const float pc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Sc), 0x138, 0xf, 0xf, true));
        const float pd = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Sd), 0x138, 0xf, 0xf, true));
        Sa = __fadd_rn(pa, a); Sb = __fadd_rn(pb, b); Sc = __fadd_rn(pc, c); Sd = __fadd_rn(pd, xd);
    }
    a = wn_lane_f(Sa, N - 1); b = wn_lane_f(Sb, N - 1); c = wn_lane_f(Sc, N - 1); d = wn_lane_f(Sd, N - 1);
}
// runtime length (n uniform, 1 <= n <= 64): the general estimator path and the whole-frame kernel
__device__ inline float wn_seq_sum_n(float x, float init, int n, int lane)
{
    const float xi = lane == 0 ? __fadd_rn(init, x) : x;
    float S = xi;
    for (int t = 1; t < n; t++)
        S = __fadd_rn(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(S), 0x138, 0xf, 0xf, true)), xi);
    return wn_lane_f(S, n - 1);
}
// Ia / s + (Jx + Jy) / (sqrt(Jx^2 + Jy^2) + 1e-20): the float quotient and sums promoted to double for the second term and the final
// addition (inpaint.cpp's expression with its double-precision sqrt)
__device__ inline double telea_estimate_d(float Ia, float Jx, float Jy, float s)
{
    return (double)__fdiv_rn(Ia, s) + (double)__fadd_rn(Jx, Jy) / (sqrt((double)__fadd_rn(__fmul_rn(Jx, Jx), __fmul_rn(Jy, Jy))) + (double)1.0e-20f);
}
__device__ inline float telea_estimate(float Ia, float Jx, float Jy, float s) { return (float)telea_estimate_d(Ia, Jx, Jy, s); }
// the uchar branch of icvTeleaInpaintFMM (8-bit images, here as floats holding 0..255): + 0.5f, cvRound, saturate_cast<uchar>
__device__ inline float telea_estimate_u8(float Ia, float Jx, float Jy, float s)
{
    const float sat = (float)(telea_estimate_d(Ia, Jx, Jy, s) + (double)0.5f);
    return fminf(fmaxf(rintf(sat), 0.f), 255.f);
}
// 1 / (1 + |T - Tc|) formed in double and rounded to float (inpaint.cpp: lev = (float)(1./(1+fabs(...))))
__device__ inline float telea_lev(float tk, float tc) { return (float)(1.0 / (1.0 + (double)fabsf(__fsub_rn(tk, tc)))); }

// Stable priority queue (T, push order), the semantics of OpenCV's CvPriorityQueueFloat.  Two sorted runs:
//   * COLD: a sorted array of (T bits << 32 | cell) words in LDS, popped at its head;
//   * HOT: the newest <= 64 pushes, sorted, one per lane in registers.  A push is a branch-free insertion
//     (one DPP shift of the lanes holding larger keys); nothing in LDS moves.
// Every hot entry is younger than every cold entry, so "cold first on equal T" is FIFO.  When the hot run is full
// it is merged into the cold array: each hot lane binary-searches its slot (after the cold entries with
// T' <= T), each cold entry above the smallest hot key binary-searches how many hot keys precede it, and every
// word moves once.  A FMM push lands ~100 entries below the tail of a single sorted array; with the buffer the
// amortised cost of a push is a handful of VALU ops plus ~1/64 of a merge.
struct WQ {
    unsigned long long *e;      // cold run, LDS [cap]
    uint32_t *hotL;             // LDS [64]: hot keys during a merge
    int head, tail, ovf;        // uniform
    int cap;                    // capacity of e (power of two)
    int nh;                     // uniform: hot entries (lanes [0, nh), ascending)
    uint32_t h0;                // uniform: smallest hot key, 0xFFFFFFFF if none
    uint32_t hk, hv;            // per lane: hot key (0xFFFFFFFF = empty) / cell
    uint32_t preT, preI;        // per lane copy of the cold head entry (valid while head < tail)
};
__device__ __attribute__((always_inline)) inline void wq_init(WQ &q)
{
    q.head = q.tail = 0; q.nh = 0; q.h0 = 0xFFFFFFFFu; q.hk = 0xFFFFFFFFu; q.hv = 0; q.preT = 0xFFFFFFFFu; q.preI = 0;
}
__device__ __attribute__((always_inline)) inline void wq_prefetch(WQ &q)
{
    // unconditional (a stale word is read when the run is empty; wq_pop checks head < tail before using it)
    unsigned long long v = q.e[q.head & (q.cap - 1)];
    q.preT = (uint32_t)(v >> 32); q.preI = (uint32_t)v;
}
// PRE: keep a prefetched copy of the cold head for wq_pop (false: the caller reads the run itself)
template <bool PRE = true>
__device__ __attribute__((always_inline)) inline void wq_merge(WQ &q, int lane)
{
    int n = q.tail - q.head;
    if (q.tail + 64 > q.cap) {
        if (q.head == 0) { q.ovf = 1; q.nh = 0; q.hk = 0xFFFFFFFFu; q.h0 = 0xFFFFFFFFu; return; }
        for (int j0 = 0; j0 < n; j0 += 64) {                // slide the cold run back to 0 (ascending chunks never clobber unread words)
            int j = j0 + lane;
            unsigned long long v = 0;
            if (j < n) v = q.e[j + q.head];
            __builtin_amdgcn_wave_barrier();
            if (j < n) q.e[j] = v;
            __builtin_amdgcn_wave_barrier();
        }
        q.head = 0; q.tail = n;
    }
    const unsigned long long *ce = q.e + q.head;
    q.hotL[lane] = q.hk;
    // slot of hot lane i: after the c cold entries with T' <= key
    int lo = 0, hi = n;
    const int it1 = n > 0 ? 32 - __builtin_clz((unsigned)n) : 0;     // ceil(log2(n + 1))
    for (int it = 0; it < it1; it++) {
        int mid = (lo + hi) >> 1;
        bool act = lo < hi;
        uint32_t tm = act ? (uint32_t)(ce[mid] >> 32) : 0u;
        if (act) { if (tm <= q.hk) lo = mid + 1; else hi = mid; }
    }
    const int c = lo;
    const int c0 = __builtin_amdgcn_readfirstlane(c);        // cold entries below c0 stay where they are
    __builtin_amdgcn_wave_barrier();
    for (int jt = n - 1; jt >= c0; jt -= 64) {               // top-down: an entry moves up by <= 64
        const int j = jt - lane;
        const bool act = j >= c0;
        const unsigned long long v = act ? ce[j] : 0ull;
        const uint32_t tj = (uint32_t)(v >> 32);
        int l2 = 0, h2 = 64;                                  // s = hot keys < tj
#pragma unroll
        for (int it = 0; it < 7; it++) {
            int mid = (l2 + h2) >> 1;
            bool a2 = l2 < h2;
            uint32_t km = q.hotL[a2 ? mid : 0];
            if (a2) { if (km < tj) l2 = mid + 1; else h2 = mid; }
        }
        __builtin_amdgcn_wave_barrier();
        if (act) q.e[q.head + j + l2] = v;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < q.nh) q.e[q.head + c + lane] = ((unsigned long long)q.hk << 32) | q.hv;      // lanes >= nh are empty (key 0xFFFFFFFF sorts last)
    __builtin_amdgcn_wave_barrier();
    q.tail += q.nh;
    q.nh = 0; q.hk = 0xFFFFFFFFu; q.h0 = 0xFFFFFFFFu;
    if (PRE) wq_prefetch(q);
}
template <bool PRE = true>
__device__ __attribute__((always_inline)) inline void wq_push(WQ &q, float Tf, int idx, int lane)
{
    if (q.nh == 64) { wq_merge<PRE>(q, lane); if (q.ovf) return; }
    const uint32_t tb = __float_as_uint(Tf);
    const uint32_t pk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q.hk, 0x138, 0xf, 0xf, false);    // wave_shr1: lane l <- l-1, lane 0 <- 0
    const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q.hv, 0x138, 0xf, 0xf, false);
    const bool gt = q.hk > tb, pgt = pk > tb;
    q.hk = gt ? (pgt ? pk : tb) : q.hk;
    q.hv = gt ? (pgt ? pv : (uint32_t)idx) : q.hv;
    q.nh++;
    q.h0 = tb < q.h0 ? tb : q.h0;
}
// pop the smallest (T, then oldest); -1 when empty (uniform)
__device__ __attribute__((always_inline)) inline int wq_pop(WQ &q)
{
    const bool cold = q.head < q.tail;
    if (!cold && q.nh == 0) return -1;
    const uint32_t cT = cold ? (uint32_t)__builtin_amdgcn_readfirstlane((int)q.preT) : 0xFFFFFFFFu;
    if (cold && cT <= q.h0) {
        int idx = __builtin_amdgcn_readfirstlane((int)q.preI);
        q.head++;
        wq_prefetch(q);
        return idx;
    }
    int idx = __builtin_amdgcn_readlane((int)q.hv, 0);
    q.hk = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)q.hk, 0x130, 0xf, 0xf, false);              // wave_shl1: lane l <- l+1
    q.hv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q.hv, 0x130, 0xf, 0xf, false);
    q.nh--;
    q.h0 = (uint32_t)__builtin_amdgcn_readlane((int)q.hk, 0);
    return idx;
}

// FMM quadrant solve (inpaint.cpp FastMarching_solve): a11 / a22 = T of the vertical / horizontal neighbour,
// k1 / k2 = that neighbour is not INSIDE
__device__ inline float wn_solve(double a11, double a22, bool k1, bool k2)
{
    // Branch-free: on a lone wave every divergent branch costs more than the arithmetic it skips.  sqrt(2 - d^2) is only used for
    // |d| < 1, i.e. an argument in (1, 2]: v_rsq_f64 and the two correction steps of the compiler's own f64 sqrt expansion, without
    // its denormal scaling -- bit-identical to sqrt() on that range.
    const double d = a11 - a22, m12 = a11 < a22 ? a11 : a22;
    const bool wide = fabs(d) >= 1.0;
    const double x = wide ? 2.0 : 2.0 - d * d;
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, hh = 0.5 * y;
    const double r = fma(-hh, g, 0.5);
    g = fma(g, r, g);
    hh = fma(hh, r, hh);
    double e = fma(-g, g, x);
    g = fma(e, hh, g);
    e = fma(-g, g, x);
    g = fma(e, hh, g);
    const double both = wide ? 1.0 + m12 : (a11 + a22 + g) * 0.5;
    const double sol = k1 ? (k2 ? both : 1.0 + a11) : (k2 ? 1.0 + a22 : 1.0 + m12);
    return (float)sol;
}

// next seed (initial band pixel) in raster order, -1 when exhausted; (base, pend) is the scan state
__device__ inline int wn_next_seed(const uint8_t *f, int cells, int &base, unsigned long long &pend, int lane)
{
    while (!pend && base < cells) {
        int li = base + lane;
        pend = __ballot(li < cells && (f[li] & W_SEED));
        if (!pend) base += 64;
    }
    if (!pend) return -1;
    int l = __ffsll((long long)pend) - 1;
    pend &= pend - 1;
    int p = base + l;
    if (!pend) base += 64;
    return p;
}

// The march on one window.  CL = false: the window holds every hole pixel of the frame (lab / rootp unused).
// CL = true: only the hole pixels of the cluster `rootp` (label plane `lab`) are INSIDE; hole pixels of other
// clusters that happen to lie in the window are treated as known pixels -- they are farther than range + 1 from
// every pixel this march reads, and they are restored by their own march.  Returns false when the queue overflowed

// ---- the two per-pop bodies of the march, shared by the frame-window and the cluster kernel -------------------------
struct TeleaWin {
    float *t, *im;          // LDS planes of the window: T field, image
    uint8_t *f;             // flag bytes
    int ww;                 // row pitch of the window
};
struct TeleaOutsideConsts { int dn, d1, d2; };       // lanes 0..15 = 4 neighbours x 4 quadrants
__device__ inline TeleaOutsideConsts telea_outside_consts(int lane, int ww)
{
    const int nb = (lane >> 2) & 3, qd = lane & 3;
    TeleaOutsideConsts c;
    c.dn = nb == 0 ? -ww : nb == 1 ? -1 : nb == 2 ? ww : 1;
    c.d1 = (qd & 1) ? ww : -ww;
    c.d2 = (qd & 2) ? 1 : -1;
    return c;
}
// this lane's neighbour offsets of the first two 64-neighbour chunks of the (2*range+1)^2 estimator window
struct TeleaMarchConsts {
    int off[2], d4, range, nn, side, r2;
    int ndisc;                  // window positions inside the disc dk^2 + dl^2 <= range^2
    float rx[2], ry[2], dstw[2];
    bool on[2];
};
__device__ inline TeleaMarchConsts telea_march_consts(int lane, int ww, int range)
{
    TeleaMarchConsts c;
    c.range = range; c.r2 = range * range; c.side = 2 * range + 1; c.nn = c.side * c.side;
    // When the disc has at most 64 positions (range <= 4) lane i takes the i-th of them in row-major order: positions outside the disc never
    // contribute, and the sequential sums then need ndisc - 1 steps instead of side^2 - 1.  Larger ranges: all side^2 positions, row-major.
    int nd = 0, mine = -1;
    for (int i = 0; i < c.nn; i++) {
        const int dk = i / c.side - range, dl = i % c.side - range;
        if (dl * dl + dk * dk <= c.r2) { if (nd == lane) mine = i; nd++; }
    }
    c.ndisc = nd;
    const bool compact = nd <= 64;
    if (compact) c.nn = nd;                 // one chunk
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) {
        int nidx = c2 * 64 + lane;
        bool valid = nidx < c.nn;
        if (compact) { valid = c2 == 0 && mine >= 0; nidx = valid ? mine : 0; }
        int dk = nidx / c.side - range, dl = nidx % c.side - range;
        c.on[c2] = valid && (dl * dl + dk * dk <= c.r2);
        c.off[c2] = c.on[c2] ? dk * ww + dl : 0;              // lanes that are off may read, unused, the centre pixel
        float ry = (float)(-dk), rx = (float)(-dl);
        c.rx[c2] = rx; c.ry[c2] = ry;
        float len2 = __fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry));
        c.dstw[c2] = len2 > 0.f ? (float)(1. / (double)__fmul_rn(len2, sqrtf(len2))) : 0.f;
    }
    c.d4 = (lane & 3) == 0 ? -ww : (lane & 3) == 1 ? -1 : (lane & 3) == 2 ? ww : 1;    // up, left, down, right
    return c;
}

// Outside T field (icvCalcFMM with `negate`): pop p (a seed of the initial band or a queue entry), give every ring
// neighbour its T and push it.  States are those of OpenCV's `out` mask: ring = INSIDE, hole and everything else KNOWN.
template <class Push>
__device__ __attribute__((always_inline)) inline void telea_pop_outside(const TeleaWin &win, const TeleaOutsideConsts &oc, int p, bool seed, int lane,
                                                                        Push push)
{
    float *t = win.t;
    uint8_t *f = win.f;
    const int dn = oc.dn, d1 = oc.d1, d2 = oc.d2;
    if (lane == 0) f[p] = (uint8_t)(seed ? (W_SEED | W_CHANGE) : W_CHANGE);   // ring pixels carry no other bit that matters
    const int pn = p + dn;
    bool ok = lane < 16 && (f[pn] & W_ST) == W_INSIDE;
    if (!__ballot(ok)) return;
    float dist = 0.f;
    if (ok) {
        const int p1 = pn + d1, p2 = pn + d2;
        float a11 = t[p1], a22 = t[p2];
        uint8_t f1 = f[p1], f2 = f[p2];
        dist = wn_solve(a11, a22, (f1 & W_ST) != W_INSIDE, (f2 & W_ST) != W_INSIDE);
    }
    float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0xB1, 0xf, 0xf, false)); dist = o < dist ? o : dist;
    o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0x4E, 0xf, 0xf, false)); dist = o < dist ? o : dist;
    for (int k = 0; k < 4; k++) {
        bool okk = __builtin_amdgcn_readlane((int)ok, k * 4) != 0;
        if (!okk) continue;
        float dk = wn_lane_f(dist, k * 4);
        int pk = __builtin_amdgcn_readlane(pn, k * 4);
        if (lane == 0) { t[pk] = dk; f[pk] = W_BAND; }
        push(dk, pk);
    }
}

// Up to FOUR pops of the outside pass at once.  A pop of this pass only uses 16 lanes (4 neighbours x 4 quadrants), so lane
// group g = lane >> 4 takes the g-th of the next entries IN QUEUE ORDER, as long as the one-at-a-time loop would have popped
// exactly these entries one after the other with the same data:
//  * a pop writes within Manhattan distance 1 of its pixel and reads within 2, so two entries at distance >= 4 do not see
//    each other's writes (telea_outside_prefix cuts the candidate list at the first pair that is closer);
//  * an entry pushed by an earlier member of the batch must not sort before a later member (T_new < T_member would put the new
//    entry first; T_new == T_member keeps the member first, FIFO): the batch is cut there before anything is written.
// Pushes go group by group, neighbour by neighbour -- the push order of the sequential loop -- so the queue contents, the FIFO
// order among equal T and every later pop are those of the sequential march: results are bit-identical.
// `cells` = the candidate cells (wave-uniform), n = how many are valid; returns the longest admissible prefix by distance.
__device__ inline int telea_outside_prefix(const int (&cells)[4], int n, int ww, uint32_t magic_ww)
{
    int r[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { r[i] = (int)__umulhi((uint32_t)cells[i], magic_ww); c[i] = cells[i] - r[i] * ww; }
    int m = 1;
#pragma unroll
    for (int j = 1; j < 4; j++) {
        bool clash = false;
#pragma unroll
        for (int i = 0; i < j; i++) { int dr = r[i] - r[j], dc = c[i] - c[j]; clash = clash || (abs(dr) + abs(dc) < 4); }
        if (m == j && j < n && !clash) m = j + 1;
    }
    return m;
}
// Where the FMM pass keeps its states: OpenCV's `out` flags in the window's flag bytes (outside pass: ring = INSIDE, hole and everything
// else KNOWN).  (A policy rather than plain code so that the pass reads the same whichever plane holds the states.)
struct FmmFlagState {
    float *t;
    uint8_t *f;
    __device__ __attribute__((always_inline)) bool inside(int c) const { return (f[c] & W_ST) == W_INSIDE; }
    __device__ __attribute__((always_inline)) void popped(int p, bool seed) const { f[p] = (uint8_t)(seed ? (W_SEED | W_CHANGE) : W_CHANGE); }   // ring pixels carry no other bit that matters
    __device__ __attribute__((always_inline)) void filled(int pn, float T, int) const { t[pn] = T; f[pn] = W_BAND; }
    __device__ __attribute__((always_inline)) void advance(int) {}
};

// cellT = the float bits of the candidates' T (all 0 for the seeds, which never wait for a push); commit(k) removes the k entries
// from the queue.  Returns the number of entries popped (1..m).
template <class State, class Commit, class Push>
__device__ __attribute__((always_inline)) inline int telea_pop_outside4(State &st, const TeleaOutsideConsts &oc, const int (&cells)[4],
                                                                        const uint32_t (&cellT)[4], int m, bool seed, int lane, Commit commit, Push push)
{
    float *t = st.t;
    const int dn = oc.dn, d1 = oc.d1, d2 = oc.d2;
    const int g = lane >> 4;
    const int p = g == 0 ? cells[0] : g == 1 ? cells[1] : g == 2 ? cells[2] : cells[3];
    const int pn = p + dn;
    const bool ok = g < m && st.inside(pn);
    unsigned long long okb = __ballot(ok);
    if (okb) {
        float dist = 0.f;
        if (ok) {
            const int p1 = pn + d1, p2 = pn + d2;
            float a11 = t[p1], a22 = t[p2];
            const bool k1 = !st.inside(p1), k2 = !st.inside(p2);           // p itself may be among them: BAND now, CHANGE after the pop -- not INSIDE either way
            dist = wn_solve(a11, a22, k1, k2);
        }
        float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0xB1, 0xf, 0xf, false)); dist = o < dist ? o : dist;
        o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0x4E, 0xf, 0xf, false)); dist = o < dist ? o : dist;
        if (m > 1) {
            // smallest T pushed by each group (row = 16 lanes = one group); T >= 0, so the float bits order like the values
            uint32_t gm = ok ? __float_as_uint(dist) : 0x7f800000u, og;
            og = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gm, 0x141, 0xf, 0xf, false); gm = og < gm ? og : gm;      // row_half_mirror
            og = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gm, 0x140, 0xf, 0xf, false); gm = og < gm ? og : gm;      // row_mirror
            uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)gm, 0);
            int mc = 1;
            if (cellT[1] <= run) {
                mc = 2;
                const uint32_t g1 = (uint32_t)__builtin_amdgcn_readlane((int)gm, 16);
                run = g1 < run ? g1 : run;
                if (m > 2 && cellT[2] <= run) {
                    mc = 3;
                    const uint32_t g2 = (uint32_t)__builtin_amdgcn_readlane((int)gm, 32);
                    run = g2 < run ? g2 : run;
                    if (m > 3 && cellT[3] <= run) mc = 4;
                }
            }
            m = mc;
            okb &= m >= 4 ? ~0ull : (1ull << (16 * m)) - 1ull;
        }
        commit(m);                            // the m entries leave the queue before their pushes enter it
        if (g < m && (lane & 15) == 0) st.popped(p, seed);
        // the new band pixels are distinct cells (four neighbours of a pop; pops of a batch are >= 4 apart): their quad leaders store
        // T and the state together; only the queue insertions are ordered
        unsigned long long pb = okb & 0x1111111111111111ull;
        if ((pb >> lane) & 1ull) st.filled(pn, dist, __popcll(pb & ((1ull << lane) - 1ull)));     // numbered in push order
        st.advance(__popcll(pb));
        // pushes in the order of the sequential loop: pop by pop, neighbour by neighbour = ascending lane among the quad leaders
        while (pb) {
            const int l = __ffsll((long long)pb) - 1;
            pb &= pb - 1ull;
            push(wn_lane_f(dist, l), __builtin_amdgcn_readlane(pn, l));
        }
    } else {
        commit(m);
        if (g < m && (lane & 15) == 0) st.popped(p, seed);
    }
    return m;
}

// One whole FMM pass on a window: the seeds (cells with W_SEED in the flag bytes, T = 0) pop first in raster order, up to four per step out
// of the current 64-cell chunk, then the queue -- the next <= 4 entries in order are the first words of the cold run once no hot key
// precedes them.  np / ns count pops / steps (diagnostics).
// the seeds of one 64-cell chunk starting at cell `base` (pend = its lanes that hold a seed), up to four per step
template <class State>
__device__ __attribute__((always_inline)) inline void telea_fmm_seed_chunk(State &st, WQ &q, const TeleaOutsideConsts &oc, int base, unsigned long long pend,
                                                                           int ww, uint32_t magic_ww, int lane, unsigned long long &np, unsigned long long &ns)
{
    while (pend && !q.ovf) {
        int cand[4];
        unsigned long long rest = pend;
#pragma unroll
        for (int k = 0; k < 4; k++) { cand[k] = base + (int)(__ffsll((long long)rest) - 1); rest &= rest - 1ull; }    // ffs(0) - 1 = -1: masked by n
        const int npend = __popcll(pend);
        const uint32_t candT[4] = {0u, 0u, 0u, 0u};
        const int m = telea_pop_outside4(st, oc, cand, candT, telea_outside_prefix(cand, npend < 4 ? npend : 4, ww, magic_ww), true, lane,
                                         [](int) {}, [&](float T_, int idx_) { wq_push<false>(q, T_, idx_, lane); });
        pend &= pend - 1ull;
        if (m > 1) pend &= pend - 1ull;
        if (m > 2) pend &= pend - 1ull;
        if (m > 3) pend &= pend - 1ull;
        np += m;
        ns++;
    }
}
// the queue of an FMM pass until it is empty: the next <= 4 entries in order are the first words of the cold run once no hot key precedes them
template <class State>
__device__ __attribute__((always_inline)) inline void telea_fmm_queue(State &st, WQ &q, const TeleaOutsideConsts &oc, int ww, uint32_t magic_ww, int lane,
                                                                      unsigned long long &np, unsigned long long &ns)
{
    while (!q.ovf) {
        int cold_n = q.tail - q.head;
        if (cold_n == 0 && q.nh == 0) break;
        const int g = lane >> 4;
        unsigned long long wv = g < cold_n ? q.e[q.head + g] : 0ull;
        const uint32_t t3 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wv >> 32), 48);
        if (q.nh > 0 && (cold_n < 4 || q.h0 < t3)) {
            wq_merge<false>(q, lane);
            cold_n = q.tail - q.head;
            wv = g < cold_n ? q.e[q.head + g] : 0ull;
        }
        if (q.ovf) break;
        const int n = cold_n < 4 ? cold_n : 4;
        int cand[4];
#pragma unroll
        for (int k = 0; k < 4; k++) cand[k] = __builtin_amdgcn_readlane((int)(uint32_t)wv, 16 * k);
        uint32_t candT[4];
#pragma unroll
        for (int k = 0; k < 4; k++) candT[k] = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wv >> 32), 16 * k);
        const int m = telea_pop_outside4(st, oc, cand, candT, telea_outside_prefix(cand, n, ww, magic_ww), false, lane,
                                         [&](int mc) { q.head += mc; }, [&](float T_, int idx_) { wq_push<false>(q, T_, idx_, lane); });
        np += m;
        ns++;
    }
}
// cell / ww == umulhi(cell, magic) for every cell with cell * ww < 2^32
__host__ __device__ inline uint32_t telea_magic_ww(int ww) { return (uint32_t)(0x100000000ull / (unsigned)ww) + 1u; }

template <class State>
__device__ __attribute__((always_inline)) inline void telea_fmm_pass(State &st, WQ &q, const uint8_t *f, int cells, int ww, int lane,
                                                                     unsigned long long &np, unsigned long long &ns)
{
    const TeleaOutsideConsts oc = telea_outside_consts(lane, ww);
    const uint32_t magic_ww = telea_magic_ww(ww);
    for (int base = 0; base < cells && !q.ovf; base += 64) {
        const int li = base + lane;
        telea_fmm_seed_chunk(st, q, oc, base, __ballot(li < cells && (f[li] & W_SEED)), ww, magic_ww, lane, np, ns);
    }
    telea_fmm_queue(st, q, oc, ww, magic_ww, lane, np, ns);
}

#ifdef VISTAF_DEBUG
__device__ unsigned long long g_fill_dbg[8];          // cycle sums of the fill's phases (frame 0 only): diagnostics
#define FSTAMP(k) do { if (blockIdx.x == 0 && lane == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_fill_dbg[k] += t_ - fstamp_; fstamp_ = t_; } } while (0)
#define FSTART() unsigned long long fstamp_ = __builtin_amdgcn_s_memtime()
#else
#define FSTAMP(k) do { } while (0)
#define FSTART() do { } while (0)
#endif

// Telea march (icvTeleaInpaintFMM): pop p, fill every 4-neighbour that is still INSIDE and push it.  Returns the number of
// pixels filled.
// NS > 0: the disc of the estimator window has NS <= 64 positions, one per lane (telea_march_consts), the case of every shipped
// configuration (range 3: 29); NS == 0: general chunk loop.
template <int NS, class Push>
__device__ __attribute__((always_inline)) inline int telea_pop_march(const TeleaWin &win, const TeleaMarchConsts &mc, int p, bool from_queue, int lane,
                                                                     Push push)
{
    float *t = win.t, *im = win.im;
    uint8_t *f = win.f;
    const int ww = win.ww, d4 = mc.d4, range = mc.range, nn = mc.nn, side = mc.side, r2 = mc.r2;
    const int *h_off = mc.off;
    const float *h_rx = mc.rx, *h_ry = mc.ry, *h_dstw = mc.dstw;
    const bool *h_on = mc.on;
    int nfill = 0;
    if (from_queue && lane == 0) f[p] = (uint8_t)(W_HOLE | W_KNOWN);
    // the four 4-neighbours, one per lane: still INSIDE?  (a neighbour's fill never changes another's flag)
    unsigned todo = (unsigned)(__ballot(lane < 4 && (f[p + d4] & W_ST) == W_INSIDE) & 0xf);
    while (todo) {
        const int qn = __ffs((int)todo) - 1;
        todo &= todo - 1;
        const int pi = p + (qn == 0 ? -ww : qn == 1 ? -1 : qn == 2 ? ww : 1);
        nfill++;
        if constexpr (NS > 0) {
            FSTART();
            // One estimator chunk (disc <= 64 positions), written as ONE basic block: on a lone wave a taken branch costs tens of cycles and a
            // dependent instruction ~9, so every LDS read is issued up front (the T word read at pi itself is the stale one, but
            // pi is INSIDE and never enters the sums), every condition is a per-lane select, and the quadrant solve (an f64 chain)
            // overlaps the image-gradient terms of the estimator.  Lanes that are off read, unused, around pi itself.
            const uint8_t f4 = f[pi + d4];
            const float t4 = t[pi + d4];
            const int pk = pi + h_off[0];
            const uint8_t f0 = f[pk], fr = f[pk + 1], fl = f[pk - 1], fd = f[pk + ww], fu = f[pk - ww];
            const float tk = t[pk];
            // OpenCV's index shifts at the first / last image row / column (km, kp, lm, lp): all 0 away from the image border
            const int sk = (fu >> 4) & 1, sK = (fd >> 4) & 1, sl = (fl >> 4) & 1, sL = (fr >> 4) & 1;
            const int rowm = pk + (sk ? ww : 0);
            const float vC = im[rowm + sl], vA = im[rowm + 1 - sL], vB = im[rowm + sl - 1], vD = im[rowm - sL];
            const float vE = im[pk + (sK ? 0 : ww) + sl], vF = im[rowm - ww + sl], vG = im[pk - (sK ? ww : 0) + sl];
            const float tu = wn_lane_f(t4, 0), tl = wn_lane_f(t4, 1), td = wn_lane_f(t4, 2), tr = wn_lane_f(t4, 3);
            // the neighbour states are wave-uniform; as scalar conditions every select below would become a branch, so they are
            // handed to the compiler as per-lane values
            unsigned kv = (unsigned)__ballot((f4 & W_ST) != W_INSIDE) & 0xf;
            asm volatile("" : "+v"(kv));
            const bool ku = kv & 1, kl = kv & 2, kd = kv & 4, kr = kv & 8;
            const int qd = lane & 3;
            float sq = wn_solve((qd & 1) ? td : tu, (qd & 2) ? tr : tl, (qd & 1) ? kd : ku, (qd & 2) ? kr : kl);
            float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), 0xB1, 0xf, 0xf, false)); sq = o < sq ? o : sq;
            o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), 0x4E, 0xf, 0xf, false)); sq = o < sq ? o : sq;
            const float tc = wn_lane_f(sq, 0);
            FSTAMP(0);                     // LDS reads + quadrant solve
            if (lane == 0) t[pi] = tc;
            // (all alternatives are evaluated, then selected: nested conditional expressions would come back as branches)
            const float gx2 = __fmul_rn(__fsub_rn(tr, tl), 0.5f), gxr = __fsub_rn(tr, tc), gxl = __fsub_rn(tc, tl);
            const float gy2 = __fmul_rn(__fsub_rn(td, tu), 0.5f), gyd = __fsub_rn(td, tc), gyu = __fsub_rn(tc, tu);
            const float gtx_r = kl ? gx2 : gxr, gtx_n = kl ? gxl : 0.f, gtx = kr ? gtx_r : gtx_n;
            const float gty_d = ku ? gy2 : gyd, gty_n = ku ? gyu : 0.f, gty = kd ? gty_d : gty_n;
            const float rx = h_rx[0], ry = h_ry[0];
            const bool use = (int)h_on[0] & (int)((f0 & W_BORDER) == 0) & (int)((f0 & W_ST) != W_INSIDE);
            const float lev = telea_lev(tk, tc);
            float dir = __fadd_rn(__fmul_rn(rx, gtx), __fmul_rn(ry, gty));
            dir = fabsf(dir) <= 0.01f ? 0.000001f : dir;          // float(0.01) < 0.01: same set of floats as the double compare
            const float wgt = fabsf(__fmul_rn(__fmul_rn(h_dstw[0], lev), dir));
            const bool nr = (fr & W_ST) != W_INSIDE, nl = (fl & W_ST) != W_INSIDE, nd = (fd & W_ST) != W_INSIDE, nu = (fu & W_ST) != W_INSIDE;
            const float ix2 = __fmul_rn(__fsub_rn(vA, vB), 2.0f), ixr = __fsub_rn(vA, vC), ixl = __fsub_rn(vD, vB);
            const float iy2 = __fmul_rn(__fsub_rn(vE, vF), 2.0f), iyd = __fsub_rn(vE, vC), iyu = __fsub_rn(vG, vF);
            const float gix_r = nl ? ix2 : ixr, gix_n = nl ? ixl : 0.f, gix = nr ? gix_r : gix_n;
            const float giy_d = nu ? iy2 : iyd, giy_n = nu ? iyu : 0.f, giy = nd ? giy_d : giy_n;
            // terms of Ia += w * I, Jx -= w * (gix * rx), Jy -= w * (giy * ry), s += w (s starts at 1e-20f); a skipped position adds 0
            const float z = 0.f;
            const float aIa = use ? __fmul_rn(wgt, vC) : z;
            const float aJx = use ? -__fmul_rn(wgt, __fmul_rn(gix, rx)) : z;
            const float aJy = use ? -__fmul_rn(wgt, __fmul_rn(giy, ry)) : z;
            const float aS = use ? wgt : z;
            float Ia = aIa, Jx = aJx, Jy = aJy, s = aS;
            FSTAMP(1);                     // weights and terms
            wn_seq_sum4<NS>(Ia, Jx, Jy, s, 1.0e-20f, lane);
            FSTAMP(2);                     // ordered sums
            const float val = telea_estimate(Ia, Jx, Jy, s);
            if (lane == 0) { im[pi] = val; f[pi] = (uint8_t)(W_HOLE | W_BAND); }
            FSTAMP(3);                     // final estimate + stores
            push(tc, pi);
            FSTAMP(4);                     // queue push
            continue;
        }
        // (flag, T) of pi's up / left / down / right neighbours on lanes 0..3, shared by all lanes
        const uint8_t f4 = f[pi + d4];
        const float t4 = t[pi + d4];
        const bool k4 = (f4 & W_ST) != W_INSIDE;
        const float tu = wn_lane_f(t4, 0), tl = wn_lane_f(t4, 1), td = wn_lane_f(t4, 2), tr = wn_lane_f(t4, 3);
        const unsigned kk = (unsigned)__ballot(k4) & 0xf;
        const bool ku = kk & 1, kl = kk & 2, kd = kk & 4, kr = kk & 8;
        float dist;
        {
            const int qd = lane & 3;      // quadrants (i-1,j-1) (i+1,j-1) (i-1,j+1) (i+1,j+1)
            float s = wn_solve((qd & 1) ? td : tu, (qd & 2) ? tr : tl, (qd & 1) ? kd : ku, (qd & 2) ? kr : kl);
            float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xf, 0xf, false)); s = o < s ? o : s;
            o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xf, 0xf, false)); s = o < s ? o : s;
            dist = wn_lane_f(s, 0);
        }
        if (lane == 0) t[pi] = dist;
        __builtin_amdgcn_wave_barrier();
        const float tc = dist;
        float gtx, gty;
        if (kr) gtx = kl ? __fmul_rn(__fsub_rn(tr, tl), 0.5f) : __fsub_rn(tr, tc);
        else gtx = kl ? __fsub_rn(tc, tl) : 0.f;
        if (kd) gty = ku ? __fmul_rn(__fsub_rn(td, tu), 0.5f) : __fsub_rn(td, tc);
        else gty = ku ? __fsub_rn(tc, tu) : 0.f;
        float Ia = 0.f, Jx = 0.f, Jy = 0.f, s = 1.0e-20f;       // running sums in OpenCV's order: chunk after chunk, lane after lane
        for (int n0 = 0; n0 < nn; n0 += 64) {
            float aIa = 0.f, aJx = 0.f, aJy = 0.f, aS = 0.f;
            int off;
            float dstw, rx, ry;
            bool on;
            if (n0 < 128) { int c2 = n0 >> 6; off = h_off[c2]; dstw = h_dstw[c2]; on = h_on[c2]; rx = h_rx[c2]; ry = h_ry[c2]; }
            else {
                int nidx = n0 + lane;
                int dk = nidx / side - range, dl = nidx % side - range;
                off = dk * ww + dl;
                on = nidx < nn && (dl * dl + dk * dk <= r2);
                ry = (float)(-dk); rx = (float)(-dl);
                float len2 = __fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry));
                dstw = len2 > 0.f ? (float)(1. / (double)__fmul_rn(len2, sqrtf(len2))) : 0.f;
            }
            if (on) {
                const int pk = pi + off;                      // inside the window: margin range+1
                const uint8_t f0 = f[pk], fr = f[pk + 1], fl = f[pk - 1], fd = f[pk + ww], fu = f[pk - ww];
                const float tk = t[pk];
                float vC = im[pk], vA = im[pk + 1], vB = im[pk - 1], vE = im[pk + ww], vF = im[pk - ww];
                if (!(f0 & W_BORDER) && (f0 & W_ST) != W_INSIDE) {
                    float vD = vC, vG = vC;
                    if ((fr | fl | fd | fu) & W_BORDER) {
                        // OpenCV's index shifts at the first / last image row / column (km, kp, lm, lp)
                        const int sk = (fu & W_BORDER) ? 1 : 0, sK = (fd & W_BORDER) ? 1 : 0;
                        const int sl = (fl & W_BORDER) ? 1 : 0, sL = (fr & W_BORDER) ? 1 : 0;
                        const int rowm = pk + sk * ww;
                        vC = im[rowm + sl]; vA = im[rowm + 1 - sL]; vB = im[rowm + sl - 1]; vD = im[rowm - sL];
                        vE = im[pk + (1 - sK) * ww + sl]; vF = im[rowm - ww + sl]; vG = im[pk - sK * ww + sl];
                    }
                    float lev = telea_lev(tk, tc);
                    float dir = __fadd_rn(__fmul_rn(rx, gtx), __fmul_rn(ry, gty));
                    if (fabsf(dir) <= 0.01f) dir = 0.000001f;   // float(0.01) < 0.01: same set of floats as the double compare
                    float wgt = fabsf(__fmul_rn(__fmul_rn(dstw, lev), dir));
                    const bool nr = (fr & W_ST) != W_INSIDE, nl = (fl & W_ST) != W_INSIDE, nd = (fd & W_ST) != W_INSIDE, nu = (fu & W_ST) != W_INSIDE;
                    float gix, giy;
                    if (nr) gix = nl ? __fmul_rn(__fsub_rn(vA, vB), 2.0f) : __fsub_rn(vA, vC);
                    else gix = nl ? __fsub_rn(vD, vB) : 0.f;
                    if (nd) giy = nu ? __fmul_rn(__fsub_rn(vE, vF), 2.0f) : __fsub_rn(vE, vC);
                    else giy = nu ? __fsub_rn(vG, vF) : 0.f;
                    aIa = __fmul_rn(wgt, vC);
                    aJx = -__fmul_rn(wgt, __fmul_rn(gix, rx));
                    aJy = -__fmul_rn(wgt, __fmul_rn(giy, ry));
                    aS = wgt;
                }
            }
            const int nl = nn - n0 < 64 ? nn - n0 : 64;
            Ia = wn_seq_sum_n(aIa, Ia, nl, lane); Jx = wn_seq_sum_n(aJx, Jx, nl, lane); Jy = wn_seq_sum_n(aJy, Jy, nl, lane); s = wn_seq_sum_n(aS, s, nl, lane);
        }
        const float val = telea_estimate(Ia, Jx, Jy, s);
        if (lane == 0) { im[pi] = val; f[pi] = (uint8_t)(W_HOLE | W_BAND); }
        push(dist, pi);
    }
    return nfill;
}

// The estimate of ONE hole pixel pi whose T the ordering pass (k_inpaint_mw.hip) has already fixed: the fill block of telea_pop_march without
// the quadrant solve and the push, same operations in the same order (so the value carries the same bits).  The caller guarantees that
// every fill within Chebyshev distance range + 1 of pi runs in march order (the block reads flags, T and image values that far out):
// then "flag != INSIDE" in the live flag bytes is exactly the flag history cv::inpaint's march would have seen at this fill.
// (T of cells that are still INSIDE already holds its final value instead of 1e6; every use of it is masked by the cell's flag.)
template <int NS>
__device__ __attribute__((always_inline)) inline void telea_fill_known_T(const TeleaWin &win, const TeleaMarchConsts &mc, int pi, int lane)
{
    static_assert(NS > 0 && NS <= 64, "one estimator chunk");
    float *t = win.t, *im = win.im;
    uint8_t *f = win.f;
    const int ww = win.ww, d4 = mc.d4;
    const uint8_t f4 = f[pi + d4];
    const float t4 = t[pi + d4];
    const float tcl = t[pi];
    const int pk = pi + mc.off[0];
    const uint8_t f0 = f[pk], fr = f[pk + 1], fl = f[pk - 1], fd = f[pk + ww], fu = f[pk - ww];
    const float tk = t[pk];
    const int sk = (fu >> 4) & 1, sK = (fd >> 4) & 1, sl = (fl >> 4) & 1, sL = (fr >> 4) & 1;
    const int rowm = pk + (sk ? ww : 0);
    const float vC = im[rowm + sl], vA = im[rowm + 1 - sL], vB = im[rowm + sl - 1], vD = im[rowm - sL];
    const float vE = im[pk + (sK ? 0 : ww) + sl], vF = im[rowm - ww + sl], vG = im[pk - (sK ? ww : 0) + sl];
    const float tu = wn_lane_f(t4, 0), tl = wn_lane_f(t4, 1), td = wn_lane_f(t4, 2), tr = wn_lane_f(t4, 3);
    const float tc = wn_lane_f(tcl, 0);
    unsigned kv = (unsigned)__ballot((f4 & W_ST) != W_INSIDE) & 0xf;
    asm volatile("" : "+v"(kv));
    const bool ku = kv & 1, kl = kv & 2, kd = kv & 4, kr = kv & 8;
    const float gx2 = __fmul_rn(__fsub_rn(tr, tl), 0.5f), gxr = __fsub_rn(tr, tc), gxl = __fsub_rn(tc, tl);
    const float gy2 = __fmul_rn(__fsub_rn(td, tu), 0.5f), gyd = __fsub_rn(td, tc), gyu = __fsub_rn(tc, tu);
    const float gtx_r = kl ? gx2 : gxr, gtx_n = kl ? gxl : 0.f, gtx = kr ? gtx_r : gtx_n;
    const float gty_d = ku ? gy2 : gyd, gty_n = ku ? gyu : 0.f, gty = kd ? gty_d : gty_n;
    const float rx = mc.rx[0], ry = mc.ry[0];
    const bool use = (int)mc.on[0] & (int)((f0 & W_BORDER) == 0) & (int)((f0 & W_ST) != W_INSIDE);
    const float lev = telea_lev(tk, tc);
    float dir = __fadd_rn(__fmul_rn(rx, gtx), __fmul_rn(ry, gty));
    dir = fabsf(dir) <= 0.01f ? 0.000001f : dir;
    const float wgt = fabsf(__fmul_rn(__fmul_rn(mc.dstw[0], lev), dir));
    const bool nr = (fr & W_ST) != W_INSIDE, nl = (fl & W_ST) != W_INSIDE, nd = (fd & W_ST) != W_INSIDE, nu = (fu & W_ST) != W_INSIDE;
    const float ix2 = __fmul_rn(__fsub_rn(vA, vB), 2.0f), ixr = __fsub_rn(vA, vC), ixl = __fsub_rn(vD, vB);
    const float iy2 = __fmul_rn(__fsub_rn(vE, vF), 2.0f), iyd = __fsub_rn(vE, vC), iyu = __fsub_rn(vG, vF);
    const float gix_r = nl ? ix2 : ixr, gix_n = nl ? ixl : 0.f, gix = nr ? gix_r : gix_n;
    const float giy_d = nu ? iy2 : iyd, giy_n = nu ? iyu : 0.f, giy = nd ? giy_d : giy_n;
    const float z = 0.f;
    float Ia = use ? __fmul_rn(wgt, vC) : z;
    float Jx = use ? -__fmul_rn(wgt, __fmul_rn(gix, rx)) : z;
    float Jy = use ? -__fmul_rn(wgt, __fmul_rn(giy, ry)) : z;
    float s = use ? wgt : z;
    wn_seq_sum4<NS>(Ia, Jx, Jy, s, 1.0e-20f, lane);
    const float val = telea_estimate(Ia, Jx, Jy, s);
    if (lane == 0) { im[pi] = val; f[pi] = (uint8_t)(W_HOLE | W_BAND); }
}

}  // namespace vf
