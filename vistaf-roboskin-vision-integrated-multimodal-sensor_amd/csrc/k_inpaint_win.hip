// cv2.inpaint(INPAINT_TELEA) -- LDS-resident window kernel (shape_ftp.py:652-666, :1199).
//
// Everything Telea's march touches lies within range+1 pixels of the hole mask: the outside T ring reaches
// `range` pixels, the FMM solves and the image gradients one pixel more.  So the exact whole-frame algorithm
// (k_inpaint.hip, OpenCV photo/inpaint.cpp restated) can run on the bounding box of the frame's hole pixels
// grown by range+1 -- and for the bad-pixel masks of this path (top 0.1 % intensity / 0.3 % gradient, dilated
// 5x5) that box is a fraction of the frame.  One wavefront per frame keeps the window's T field, image and
// flag plane plus the priority queue in LDS (<= 160 KB): every dependent read of the march is an LDS access
// instead of an L2 round trip.  Raster order inside a rectangle equals raster order in the frame, so seeds
// and queue ties pop in the same order as in the whole-frame kernel; T, the march order and therefore the
// set of neighbours every estimate uses are identical (the float sums are reduced in a different order).
// Frames whose window or queue does not fit raise fb[b]; the whole-frame kernel then handles exactly those.
//
// The march is a chain of dependent steps on ONE wave (about 8 cycles per dependent VALU op, 64+ per LDS
// round trip), so the code is organised to minimise dependent steps, not work:
//   * the window is NOT clipped to the frame: cells beyond the image carry a BORDER flag, so the inner
//     loops need no coordinates, divisions or bounds checks -- `pi + offset` is always a valid cell and
//     OpenCV's first/last row/column index shifts are read off the neighbours' BORDER bits;
//   * the four 4-neighbours of a popped pixel are tested by four lanes in one LDS round;
//   * the neighbours' (flag, T) pairs loaded for the FMM solve are reused for grad T;
//   * queue entries are packed (T bits << 32 | cell) so a push moves one 64-bit word per lane;
//   * the (2r+1)^2 estimator terms are accumulated per lane over the 64-neighbour chunks and reduced once.
#include <cstdio>
#include "kernels.hpp"

namespace vf {

constexpr int WN_QCAP = 4096;       // live queue entries (8 B each)
constexpr int WN_CELLS = 14464;     // window cells (9 B each: T f32, image f32, flags u8)
// flag byte: bits 0-1 state, bit 4 outside the image, bit 5 scratch (hole in row range), bit 6 hole pixel,
// bit 7 seed (initial band)
constexpr uint8_t W_KNOWN = 0, W_BAND = 1, W_INSIDE = 2, W_CHANGE = 3, W_ST = 3, W_BORDER = 0x10, W_ROW = 0x20, W_HOLE = 0x40, W_SEED = 0x80;

__device__ unsigned long long g_win_dbg[1024][16];   // diagnostic shader-clock stamps per frame (VISTAF_TELEA_DBG)
#define WSTAMP(i) do { if (!CL && lane == 0 && b < 1024) g_win_dbg[b][i] = __builtin_amdgcn_s_memtime(); } while (0)

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

// box planes [4][B]: xmin, ymin (start 0x7f7f7f7f), xmax, ymax (start 0) of the hole pixels of each frame
__global__ void k_bad_bbox(const uint8_t *__restrict__ bad, int32_t *__restrict__ box, int B, int h, int w)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    int P = h * w;
    bool on = p < P && bad[(size_t)b * P + p];
    if (!__ballot(on)) return;
    int y = p / w, x = p - y * w;
    int x0 = on ? x : 0x7f7f7f7f, y0 = on ? y : 0x7f7f7f7f, x1 = on ? x : 0, y1 = on ? y : 0;
    for (int o = 32; o; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&box[b], x0); atomicMin(&box[B + b], y0); atomicMax(&box[2 * B + b], x1); atomicMax(&box[3 * B + b], y1);
    }
}

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
    q.e[q.head + c + lane] = ((unsigned long long)q.hk << 32) | q.hv;
    __builtin_amdgcn_wave_barrier();
    q.tail += 64;
    q.nh = 0; q.hk = 0xFFFFFFFFu; q.h0 = 0xFFFFFFFFu;
    wq_prefetch(q);
}
__device__ __attribute__((always_inline)) inline void wq_push(WQ &q, float Tf, int idx, int lane)
{
    if (q.nh == 64) { wq_merge(q, lane); if (q.ovf) return; }
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
    double m12 = a11 < a22 ? a11 : a22;
    double sol;
    if (k1) {
        if (k2) {
            if (fabs(a11 - a22) >= 1.0) sol = 1 + m12;
            else sol = (a11 + a22 + sqrt((double)(2 - (a11 - a22) * (a11 - a22)))) * 0.5;
        } else sol = 1 + a11;
    } else if (k2) sol = 1 + a22;
    else sol = 1 + m12;
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
// (nothing has been written back).
template <bool CL>
__device__ inline bool wn_march(float *__restrict__ img, const uint8_t *__restrict__ bad, const int32_t *__restrict__ lab, int rootp,
                                int i0, int j0, int wh, int ww, int range, int h, int w, unsigned char *lds, int cells_cap, int qcap,
                                int lane, int b)
{
    unsigned long long *qe = (unsigned long long *)lds;          // [qcap]
    float *t = (float *)(qe + qcap);                             // [cells_cap]
    float *im = t + cells_cap;                                   // [cells_cap]
    uint8_t *f = (uint8_t *)(im + cells_cap);                    // [cells_cap]
    uint32_t *hotL = (uint32_t *)(f + cells_cap);                // [64]
    const int cells = wh * ww;
    WSTAMP(0);

    // ---- load window: hole / border bits, T = 1e6, image
    for (int r = 0; r < wh; r++) {
        const int gi = i0 + r;
        const bool rowin = gi >= 1 && gi <= h;
#pragma unroll 2
        for (int cc = lane; cc < ww; cc += 64) {
            int gj = j0 + cc;
            bool interior = rowin && gj >= 1 && gj <= w;
            size_t gp = interior ? (size_t)(gi - 1) * w + (gj - 1) : 0;
            float v = img[gp];
            uint8_t bd = bad[gp];
            if (CL) bd = (bd && lab[gp] == rootp) ? 1 : 0;
            int li = r * ww + cc;
            im[li] = interior ? v : 0.f;
            f[li] = !interior ? W_BORDER : bd ? W_HOLE : (uint8_t)0;
            t[li] = 1.0e6f;
        }
    }
    __builtin_amdgcn_wave_barrier();
    WSTAMP(1);
    // ---- ring = within Chebyshev `range` of the hole (separable: rows, then columns)
    for (int r = 0; r < wh; r++)
        for (int cc = lane; cc < ww; cc += 64) {
            int li = r * ww + cc;
            int lo = max(0, cc - range), hi = min(ww - 1, cc + range);
            uint8_t any = 0;
            for (int c2 = lo; c2 <= hi; c2++) any |= f[r * ww + c2];
            if (any & W_HOLE) f[li] |= W_ROW;      // neighbours only look at bit 6, which this never changes
        }
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < wh; r++)
        for (int cc = lane; cc < ww; cc += 64) {
            int li = r * ww + cc;
            uint8_t me = f[li];
            if (me & (W_BORDER | W_HOLE)) continue;                            // hole pixels are KNOWN for the outside pass
            // window edge cells are at distance range+1 from the hole: never band, their neighbours are not needed
            bool band = (cc > 0 && (f[li - 1] & W_HOLE)) || (cc < ww - 1 && (f[li + 1] & W_HOLE)) ||
                        (r > 0 && (f[li - ww] & W_HOLE)) || (r < wh - 1 && (f[li + ww] & W_HOLE));
            if (band) { f[li] = me | W_SEED; t[li] = 0.f; continue; }
            int lo = max(0, r - range), hi = min(wh - 1, r + range);
            uint8_t any = 0;
            for (int r2 = lo; r2 <= hi; r2++) any |= f[r2 * ww + cc];
            if (any & W_ROW) f[li] = me | W_INSIDE;                              // writes bits 0-1 only
        }
    __builtin_amdgcn_wave_barrier();

    WSTAMP(2);
    WQ q;
    q.e = qe; q.hotL = hotL; q.ovf = 0; q.cap = qcap;
    wq_init(q);
    unsigned long long np1 = 0, np2 = 0, nfill = 0;
    // ---- pass 1: outside T field (icvCalcFMM, negate); seeds pop first in raster order, then the queue.
    // States here are those of OpenCV's `out` mask: ring = INSIDE, hole and everything else KNOWN.
    {
        // lanes 0..15 = 4 neighbours x 4 quadrants
        const int nb = (lane >> 2) & 3, qd = lane & 3;
        const int dn = nb == 0 ? -ww : nb == 1 ? -1 : nb == 2 ? ww : 1;
        const int d1 = (qd & 1) ? ww : -ww, d2 = (qd & 2) ? 1 : -1;
        for (int phase = 0; phase < 2 && !q.ovf; phase++) {
            if (phase == 1) WSTAMP(3);
            int base = 0;
            unsigned long long pend = 0;
            for (;;) {
                int p = phase == 0 ? wn_next_seed(f, cells, base, pend, lane) : wq_pop(q);
                if (p < 0) break;
                np1++;
                if (lane == 0) f[p] = (uint8_t)(phase == 0 ? (W_SEED | W_CHANGE) : W_CHANGE);   // ring pixels carry no other bit that matters
                const int pn = p + dn;
                bool ok = lane < 16 && (f[pn] & W_ST) == W_INSIDE;
                if (!__ballot(ok)) continue;
                float dist = 0.f;
                if (ok) {
                    const int p1 = pn + d1, p2 = pn + d2;
                    float a11 = t[p1], a22 = t[p2];
                    uint8_t f1 = f[p1], f2 = f[p2];
                    dist = wn_solve(a11, a22, (f1 & W_ST) != W_INSIDE, (f2 & W_ST) != W_INSIDE);
                }
                float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0xB1, 0xf, 0xf, false)); dist = o < dist ? o : dist;
                o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0x4E, 0xf, 0xf, false)); dist = o < dist ? o : dist;
#pragma nounroll
                for (int k = 0; k < 4; k++) {
                    bool okk = __builtin_amdgcn_readlane((int)ok, k * 4) != 0;
                    if (!okk) continue;
                    float dk = wn_lane_f(dist, k * 4);
                    int pk = __builtin_amdgcn_readlane(pn, k * 4);
                    if (lane == 0) { t[pk] = dk; f[pk] = W_BAND; }
                    wq_push(q, dk, pk, lane);
                }
            }
        }
    }
    WSTAMP(4);
    // negate T where the outside pass ran; switch the state bits to the march's flags (hole = INSIDE, rest KNOWN)
    for (int li = lane; li < cells; li += 64) {
        uint8_t v = f[li];
        if ((v & W_ST) == W_CHANGE) t[li] = -t[li];
        f[li] = (uint8_t)((v & (W_SEED | W_HOLE | W_BORDER)) | ((v & W_HOLE) ? W_INSIDE : W_KNOWN));
    }
    __builtin_amdgcn_wave_barrier();

    // ---- pass 2: Telea march (icvTeleaInpaintFMM)
    wq_init(q);
    const int r2 = range * range, side = 2 * range + 1, nn = side * side;
    // this lane's neighbour offsets of the first two 64-neighbour chunks, hoisted out of the march
    int h_off[2];
    float h_rx[2], h_ry[2], h_dstw[2];
    bool h_on[2];
    for (int c2 = 0; c2 < 2; c2++) {
        int nidx = c2 * 64 + lane;
        int dk = nidx / side - range, dl = nidx % side - range;
        h_off[c2] = dk * ww + dl;
        h_on[c2] = nidx < nn && (dl * dl + dk * dk <= r2);
        float ry = (float)(-dk), rx = (float)(-dl);
        h_rx[c2] = rx; h_ry[c2] = ry;
        float len2 = __fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry));
        h_dstw[c2] = len2 > 0.f ? (float)(1. / (double)__fmul_rn(len2, __fsqrt_rn(len2))) : 0.f;
    }
    const int d4 = (lane & 3) == 0 ? -ww : (lane & 3) == 1 ? -1 : (lane & 3) == 2 ? ww : 1;    // up, left, down, right
    WSTAMP(5);
    for (int phase = 0; phase < 2 && !q.ovf; phase++) {
        if (phase == 1) WSTAMP(6);
        int base = 0;
        unsigned long long pend = 0;
        for (;;) {
            int p = phase == 0 ? wn_next_seed(f, cells, base, pend, lane) : wq_pop(q);
            if (p < 0) break;
            np2++;
            if (phase == 1 && lane == 0) f[p] = (uint8_t)(W_HOLE | W_KNOWN);
            // the four 4-neighbours, one per lane: still INSIDE?  (a neighbour's fill never changes another's flag)
            unsigned todo = (unsigned)(__ballot(lane < 4 && (f[p + d4] & W_ST) == W_INSIDE) & 0xf);
            while (todo) {
                const int qn = __ffs((int)todo) - 1;
                todo &= todo - 1;
                const int pi = p + (qn == 0 ? -ww : qn == 1 ? -1 : qn == 2 ? ww : 1);
                nfill++;
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
                float aIa = 0.f, aJx = 0.f, aJy = 0.f, aS = 0.f;
                for (int n0 = 0; n0 < nn; n0 += 64) {
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
                        dstw = len2 > 0.f ? (float)(1. / (double)__fmul_rn(len2, __fsqrt_rn(len2))) : 0.f;
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
                            // 1 / (1 + |T - Tc|): OpenCV forms it in double and rounds to float; the float quotient differs
                            // from that by at most one ulp of a weight, far inside the estimator's own rounding noise
                            float lev = __fdiv_rn(1.0f, __fadd_rn(1.0f, fabsf(__fsub_rn(tk, tc))));
                            float dir = __fadd_rn(__fmul_rn(rx, gtx), __fmul_rn(ry, gty));
                            if (fabsf(dir) <= 0.01f) dir = 0.000001f;   // float(0.01) < 0.01: same set of floats as the double compare
                            float wgt = fabsf(__fmul_rn(__fmul_rn(dstw, lev), dir));
                            const bool nr = (fr & W_ST) != W_INSIDE, nl = (fl & W_ST) != W_INSIDE, nd = (fd & W_ST) != W_INSIDE, nu = (fu & W_ST) != W_INSIDE;
                            float gix, giy;
                            if (nr) gix = nl ? __fmul_rn(__fsub_rn(vA, vB), 2.0f) : __fsub_rn(vA, vC);
                            else gix = nl ? __fsub_rn(vD, vB) : 0.f;
                            if (nd) giy = nu ? __fmul_rn(__fsub_rn(vE, vF), 2.0f) : __fsub_rn(vE, vC);
                            else giy = nu ? __fsub_rn(vG, vF) : 0.f;
                            aIa = __fadd_rn(aIa, __fmul_rn(wgt, vC));
                            aJx = __fsub_rn(aJx, __fmul_rn(wgt, __fmul_rn(gix, rx)));
                            aJy = __fsub_rn(aJy, __fmul_rn(wgt, __fmul_rn(giy, ry)));
                            aS = __fadd_rn(aS, wgt);
                        }
                    }
                }
                const float Ia = wn_dpp_sum(aIa), Jx = wn_dpp_sum(aJx), Jy = wn_dpp_sum(aJy), s = __fadd_rn(wn_dpp_sum(aS), 1.0e-20f);
                // Ia/s + (Jx+Jy)/(sqrt(Jx^2+Jy^2) + 1e-20): the second term is a ratio in [-sqrt2, sqrt2]; OpenCV forms it in
                // double, its float evaluation differs by < 2e-7 absolute before the final rounding to float
                const float nrm = __fadd_rn(__fsqrt_rn(__fadd_rn(__fmul_rn(Jx, Jx), __fmul_rn(Jy, Jy))), 1.0e-20f);
                const float val = __fadd_rn(__fdiv_rn(Ia, s), __fdiv_rn(__fadd_rn(Jx, Jy), nrm));
                if (lane == 0) { im[pi] = val; f[pi] = (uint8_t)(W_HOLE | W_BAND); }
                wq_push(q, dist, pi, lane);
            }
        }
    }
    WSTAMP(7);
    if (!CL && lane == 0 && b < 1024) { g_win_dbg[b][8] = (np1 << 32) | np2; g_win_dbg[b][9] = (nfill << 32) | (unsigned)cells; }
    if (q.ovf) return false;
    // ---- write back the hole pixels
    for (int r = 0; r < wh; r++) {
        const int gi = i0 + r;
        for (int cc = lane; cc < ww; cc += 64) {
            int li = r * ww + cc;
            if (f[li] & W_HOLE) img[(size_t)(gi - 1) * w + (j0 + cc - 1)] = im[li];
        }
    }
    return true;
}

__global__ __launch_bounds__(64) void k_telea_window(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all,
                                                     const int32_t *__restrict__ box, int32_t *__restrict__ fb, int range, int B, int h, int w)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wn_lds[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int P = h * w;
    const int xmin = box[b], ymin = box[B + b], xmax = box[2 * B + b], ymax = box[3 * B + b];
    if (xmin == 0x7f7f7f7f) return;                              // no hole pixel: nothing to inpaint
    const int M = range + 1;
    // window in padded frame coordinates [i0, i1] x [j0, j1], not clipped (cells beyond the image: BORDER)
    const int i0 = ymin + 1 - M, i1 = ymax + 1 + M;
    const int j0 = xmin + 1 - M, j1 = xmax + 1 + M;
    const int wh = i1 - i0 + 1, ww = j1 - j0 + 1;
    bool ok = wh * ww <= WN_CELLS;
    if (ok) ok = wn_march<false>(img_all + (size_t)b * P, bad_all + (size_t)b * P, nullptr, 0, i0, j0, wh, ww, range, h, w, wn_lds, WN_CELLS, WN_QCAP, lane, b);
    if (!ok && lane == 0) fb[b] = 1;
}

// Cluster front end: hole pixels farther apart than 2 * range + 3 never interact (rings and neighbourhoods reach
// range + 1), so every connected component of the hole mask dilated by a (2 * (range + 1) + 1)^2 square -- a CLUSTER --
// can be marched on its own small window with its own queue, in exactly the order the whole-frame queue would
// pop its pixels.  One wave per cluster, several clusters per CU: the batch turns from B sequential marches into
// thousands of short ones.  Clusters whose window or queue does not fit are flagged in big[] and left to
// k_telea_window, which then sees only their pixels.
constexpr int CL2_CELLS = 3072;     // window cells per cluster
constexpr int CL2_QCAP = 512;       // queue entries per cluster
constexpr int CL2_SLOTS = 48;       // workgroups per frame (each walks clusters c, c + CL2_SLOTS, ...)

__global__ __launch_bounds__(64) void k_telea_clusters2(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all,
                                                        const int32_t *__restrict__ labels_all, const int32_t *__restrict__ list_all,
                                                        const int32_t *__restrict__ count, const int32_t *__restrict__ xmin,
                                                        const int32_t *__restrict__ ymin, const int32_t *__restrict__ xmax,
                                                        const int32_t *__restrict__ ymax, uint8_t *__restrict__ big, int range, int h, int w)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wn_lds[];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.y;
    const int P = h * w;
    const int ncl = count[b];
    const int M = range + 1;
    for (int c = blockIdx.x; c < ncl; c += gridDim.x) {
        const int rootp = list_all[b * (size_t)P + c];
        const size_t root = b * (size_t)P + rootp;
        const int i0 = ymin[root] + 1 - M, i1 = ymax[root] + 1 + M;
        const int j0 = xmin[root] + 1 - M, j1 = xmax[root] + 1 + M;
        const int wh = i1 - i0 + 1, ww = j1 - j0 + 1;
        bool ok = wh * ww <= CL2_CELLS;
        if (ok) ok = wn_march<true>(img_all + b * (size_t)P, bad_all + b * (size_t)P, labels_all + b * (size_t)P, rootp, i0, j0, wh, ww, range, h, w,
                                    wn_lds, CL2_CELLS, CL2_QCAP, lane, 0);
        if (!ok && lane == 0) big[root] = 1;
        __builtin_amdgcn_wave_barrier();
    }
}

int inpaint_cluster_cells_cap() { return CL2_CELLS; }

void launch_telea_clusters2(float *img, const uint8_t *bad, const int32_t *labels, const int32_t *list, const int32_t *count, const int32_t *xmin,
                            const int32_t *ymin, const int32_t *xmax, const int32_t *ymax, uint8_t *big, int range, int B, int h, int w, hipStream_t st)
{
    const size_t lds = (size_t)CL2_CELLS * 9 + (size_t)CL2_QCAP * 8 + 256;
    hipLaunchKernelGGL(k_telea_clusters2, dim3(CL2_SLOTS, B), dim3(64), lds, st, img, bad, labels, list, count, xmin, ymin, xmax, ymax, big, range, h, w);
}

void telea_window_debug_dump(int B)
{
    static unsigned long long hbuf[1024][16];
    if (hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_win_dbg), sizeof(hbuf)) != hipSuccess) return;
    int worst = 0;
    double mean = 0;
    for (int b = 0; b < B && b < 1024; b++) {
        if (hbuf[b][7] - hbuf[b][0] > hbuf[worst][7] - hbuf[worst][0]) worst = b;
        mean += (double)(hbuf[b][7] - hbuf[b][0]) / B;
    }
    for (int b : {0, worst}) {
        unsigned long long *x = hbuf[b];
        printf("[telea window dbg] frame %d cycles: load %llu | ring %llu | p1 seeds %llu | p1 queue %llu | negate %llu | p2 seeds %llu | p2 queue %llu | "
               "pops %llu / %llu | filled %llu | cells %llu | mean total over frames %.0f\n", b, x[1] - x[0], x[2] - x[1], x[3] - x[2], x[4] - x[3], x[5] - x[4], x[6] - x[5], x[7] - x[6],
               x[8] >> 32, x[8] & 0xffffffffull, x[9] >> 32, x[9] & 0xffffffffull, mean);
    }
}

size_t inpaint_win_scratch_bytes(int B) { return (size_t)B * 5 * sizeof(int32_t) + 256; }

// box scratch: [4][B] bbox planes + [B] fallback flags.  Returns the device pointer of the fallback flags.
int32_t *launch_inpaint_window(float *img, const uint8_t *bad, int range, void *scratch, int B, int h, int w, hipStream_t st, hipEvent_t ev_march)
{
    int32_t *box = (int32_t *)scratch, *fb = box + 4 * (size_t)B;
    (void)hipMemsetAsync(box, 0x7f, (size_t)B * 8, st);
    (void)hipMemsetAsync(box + 2 * (size_t)B, 0, (size_t)B * 12, st);
    const int P = h * w;
    hipLaunchKernelGGL(k_bad_bbox, dim3((P + 255) / 256, B), dim3(256), 0, st, bad, box, B, h, w);
    const size_t lds = (size_t)WN_CELLS * 9 + (size_t)WN_QCAP * 8 + 256;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)k_telea_window, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    if (ev_march) (void)hipEventRecord(ev_march, st);     // stage timing: the march starts here (the bbox pass belongs to the mask stage)
    hipLaunchKernelGGL(k_telea_window, dim3(B), dim3(64), lds, st, img, bad, box, fb, range, B, h, w);
    return fb;
}

}  // namespace vf
