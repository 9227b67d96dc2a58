// cv2.inpaint(INPAINT_TELEA) -- cluster-parallel front end (shape_ftp.py:652-666).
//
// Telea's march is ordered by a stable priority queue (T, push sequence), but hole pixels farther apart
// than their (2*range+1)^2 neighbourhoods plus the outside T ring can never see each other: restricted to
// such an independent CLUSTER of hole pixels the queue pops in exactly the same relative order.  So:
//   1. dilate the hole mask by a (2*(range+1)+1)^2 square and label its 8-connected components
//      (k_morph + k_cc_label): hole pixels of different components are > 2*range+3 apart (Chebyshev),
//      while rings / neighbourhoods only reach across gaps of at most 2*range+1;
//   2. k_cluster_bbox / k_cluster_list: bounding box of the hole pixels of every component, list of
//      components per frame; components whose window (bbox + range+1) exceeds CL_WMAX cells are left to
//      the sequential whole-frame kernel (k_inpaint.hip) through a second, disjoint hole mask;
//   3. k_telea_clusters: waves pull clusters from the per-frame list; a wave copies the cluster's window
//      (flags, T, image) into LDS and runs the exact OpenCV algorithm there (outside T field pass, then the
//      Telea march), one neighbour per lane for the (2r+1)^2 estimator sums, and writes back only the
//      cluster's own hole pixels.
#include "kernels.hpp"

namespace vf {

constexpr int CL_WMAX = 8192;       // window cells per cluster handled in LDS (16 bytes per cell)
constexpr int CL_WAVES = 1;         // waves per workgroup (each wave owns a full window)
constexpr uint8_t C_KNOWN = 0, C_BAND = 1, C_INSIDE = 2, C_CHANGE = 3, C_SEED = 0x80;

__device__ inline uint32_t cl_dpp_min_u32(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// full-wave float sum (DPP butterfly; result uniform)
__device__ inline float cl_dpp_sum_f32(float x)
{
    int v = __float_as_int(x);
#define VF_ADD(ctrl, rm, keep)                                                                          \
    v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(keep, v, ctrl, rm, 0xf, false)));
    VF_ADD(0xB1, 0xf, 0) VF_ADD(0x4E, 0xf, 0) VF_ADD(0x141, 0xf, 0) VF_ADD(0x140, 0xf, 0) VF_ADD(0x142, 0xa, 0) VF_ADD(0x143, 0xc, 0)
#undef VF_ADD
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}

// ---- 2. cluster bookkeeping ---------------------------------------------------------------------------
// box planes (int32 [B,P], indexed by component root): xmin / ymin start at 0x7f7f7f7f, xmax / ymax at 0
__global__ void k_cluster_bbox(const uint8_t *__restrict__ bad, const int32_t *__restrict__ labels, int32_t *xmin, int32_t *ymin,
                               int32_t *xmax, int32_t *ymax, int h, int w)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    int P = h * w;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    if (!bad[i]) return;
    size_t r = b * (size_t)P + labels[i];
    int y = p / w, x = p - y * w;
    atomicMin(&xmin[r], x); atomicMin(&ymin[r], y); atomicMax(&xmax[r], x); atomicMax(&ymax[r], y);
}

// list[b*P + k] = root of the k-th small cluster of frame b; big[root] = 1 for clusters left to the
// sequential kernel
__global__ void k_cluster_list(const int32_t *__restrict__ labels, const int32_t *__restrict__ xmin, const int32_t *__restrict__ ymin,
                               const int32_t *__restrict__ xmax, const int32_t *__restrict__ ymax, int32_t *__restrict__ list,
                               int32_t *__restrict__ count, uint8_t *__restrict__ big, int range, int h, int w)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    int P = h * w;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    if (labels[i] != p || xmin[i] == 0x7f7f7f7f) return;
    int M = range + 1;
    int i0 = max(0, ymin[i] + 1 - M), i1 = min(h + 1, ymax[i] + 1 + M);
    int j0 = max(0, xmin[i] + 1 - M), j1 = min(w + 1, xmax[i] + 1 + M);
    int cells = (i1 - i0 + 1) * (j1 - j0 + 1);
    bool isbig = cells > CL_WMAX;
    big[i] = isbig ? 1 : 0;
    if (!isbig) { int k = atomicAdd(&count[b], 1); list[b * (size_t)P + k] = p; }
}

// bad_big = bad & big[root]  (second hole mask for the sequential kernel)
__global__ void k_split_bad(const uint8_t *__restrict__ bad, const int32_t *__restrict__ labels, const uint8_t *__restrict__ big,
                            uint8_t *__restrict__ bad_big, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    bad_big[i] = (uint8_t)(bad[i] && big[b * (size_t)P + labels[i]]);
}

// ---- 3. per-cluster Telea in LDS ------------------------------------------------------------------------
// Stable priority queue kept as a SORTED array (the structure of OpenCV's CvPriorityQueueFloat): FMM pushes are
// nearly monotone in T, so an insertion only shifts the few trailing entries with a larger T (found and
// moved by the 64 lanes at once); a new entry goes AFTER every entry with T' <= T (FIFO among ties); pop is
// the head.  No sequence numbers are needed.  Capacity = pushes per pass <= window cells.
struct ClQ {
    uint32_t *T;       // float bits of T (>= 0), ascending in [head, tail)
    uint16_t *idx;
    int head, tail;
};
__device__ inline void clq_push(ClQ &q, float Tf, int idx, int lane)
{
    const uint32_t tb = __float_as_uint(Tf);
    int k = 0;
    for (;;) {
        int j = q.tail - 1 - k - lane;
        bool in = j >= q.head;
        uint32_t tv = in ? q.T[j] : 0u;
        uint16_t iv = in ? q.idx[j] : (uint16_t)0;
        unsigned long long g = __ballot(in && tv > tb);
        int c = (g == ~0ull) ? 64 : (int)(__ffsll((long long)~g) - 1);     // leading run of "greater" entries
        if (lane < c) { q.T[j + 1] = tv; q.idx[j + 1] = iv; }
        k += c;
        if (c < 64) break;
    }
    if (lane == 0) { q.T[q.tail - k] = tb; q.idx[q.tail - k] = (uint16_t)idx; }
    q.tail++;
}
// pop the smallest (T, then oldest); -1 when empty
__device__ inline int clq_pop(ClQ &q, int lane)
{
    (void)lane;
    if (q.head == q.tail) return -1;
    int idx = q.idx[q.head];
    q.head++;
    return __builtin_amdgcn_readfirstlane(idx);
}

__device__ inline float cl_solve(const uint8_t *f, const float *t, int p1, int p2)
{
    double a11 = t[p1], a22 = t[p2];
    double m12 = a11 < a22 ? a11 : a22;
    bool k1 = (f[p1] & 0x7f) != C_INSIDE, k2 = (f[p2] & 0x7f) != C_INSIDE;
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

__global__ __launch_bounds__(64 * CL_WAVES) void k_telea_clusters(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all,
                                                                  const int32_t *__restrict__ labels_all, const int32_t *__restrict__ list_all, const int32_t *__restrict__ count,
                                                                  int32_t *__restrict__ cursor, const int32_t *__restrict__ xmin,
                                                                  const int32_t *__restrict__ ymin, const int32_t *__restrict__ xmax,
                                                                  const int32_t *__restrict__ ymax, int range, int h, int w)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_lds[];
    float *s_t = (float *)cl_lds;                                   // [CL_WMAX]
    float *s_im = s_t + CL_WMAX;                                    // [CL_WMAX]
    uint32_t *s_qT = (uint32_t *)(s_im + CL_WMAX);                  // [CL_WMAX]
    uint16_t *s_qi = (uint16_t *)(s_qT + CL_WMAX);                  // [CL_WMAX]
    uint8_t *s_f = (uint8_t *)(s_qi + CL_WMAX);                     // [CL_WMAX]
    uint8_t *s_fo = s_f + CL_WMAX;                                  // [CL_WMAX]
    const int lane = threadIdx.x & 63;
    const size_t b = blockIdx.y;
    const int P = h * w, er = h + 2, ec = w + 2;
    float *img = img_all + b * (size_t)P;
    const uint8_t *bad = bad_all + b * (size_t)P;
    const int32_t *lab = labels_all + b * (size_t)P;
    const int32_t *list = list_all + b * (size_t)P;
    const int ncl = count[b];
    float *t = s_t, *im = s_im;
    uint8_t *f = s_f, *fo = s_fo;
    ClQ q;
    q.T = s_qT; q.idx = s_qi; q.head = q.tail = 0;
    const int r2 = range * range, side = 2 * range + 1, nn = side * side, M = range + 1;

    for (;;) {
        int c = 0;
        if (lane == 0) c = atomicAdd(&cursor[b], 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= ncl) break;
        const int rootp = list[c];
        const size_t root = b * (size_t)P + rootp;
        // window in padded frame coordinates [i0, i1] x [j0, j1]
        const int i0 = max(0, ymin[root] + 1 - M), i1 = min(h + 1, ymax[root] + 1 + M);
        const int j0 = max(0, xmin[root] + 1 - M), j1 = min(w + 1, xmax[root] + 1 + M);
        const int wh = i1 - i0 + 1, ww = j1 - j0 + 1, cells = wh * ww;

        // ---- load window: flags (INSIDE on the hole), T = 1e6, image
        for (int li = lane; li < cells; li += 64) {
            int r = li / ww, cc = li - r * ww;
            int gi = i0 + r, gj = j0 + cc;
            bool interior = gi >= 1 && gi <= h && gj >= 1 && gj <= w;
            size_t gp = interior ? (size_t)(gi - 1) * w + (gj - 1) : 0;
            im[li] = interior ? img[gp] : 0.f;
            // a bounding box can contain hole pixels of OTHER (independent) clusters: they stay known here
            f[li] = (interior && bad[gp] && lab[gp] == rootp) ? C_INSIDE : C_KNOWN;
            t[li] = 1.0e6f;
        }
        // ---- band seeds (T = 0) and the outside ring
        for (int li = lane; li < cells; li += 64) {
            int r = li / ww, cc = li - r * ww;
            int gi = i0 + r, gj = j0 + cc;
            bool interior = gi >= 1 && gi <= h && gj >= 1 && gj <= w;
            uint8_t v = C_KNOWN;
            if (interior && f[li] != C_INSIDE) {
                bool band = (cc > 0 && f[li - 1] == C_INSIDE) || (cc < ww - 1 && f[li + 1] == C_INSIDE) ||
                            (r > 0 && f[li - ww] == C_INSIDE) || (r < wh - 1 && f[li + ww] == C_INSIDE);
                if (band) { v = C_KNOWN | C_SEED; t[li] = 0.f; }
                else {
                    bool near = false;
                    for (int a = -range; a <= range && !near; a++) {
                        int rr = r + a; if (rr < 0 || rr >= wh) continue;
                        for (int d = -range; d <= range; d++) {
                            int c2 = cc + d; if (c2 < 0 || c2 >= ww) continue;
                            if (f[rr * ww + c2] == C_INSIDE) { near = true; break; }
                        }
                    }
                    if (near) v = C_INSIDE;
                }
            }
            fo[li] = v;
        }

        // ---- pass 1: outside T field (icvCalcFMM with negate); seeds pop first, in raster order
        q.head = q.tail = 0;
        for (int phase = 0; phase < 2; phase++) {
            int base = 0;
            unsigned long long pend = 0;
            for (;;) {
                int p;
                if (phase == 0) {
                    while (!pend && base < cells) {
                        int li = base + lane;
                        pend = __ballot(li < cells && (fo[li] & C_SEED));
                        if (!pend) base += 64;
                    }
                    if (!pend) break;
                    int l = __ffsll((long long)pend) - 1;
                    pend &= pend - 1;
                    p = base + l;
                    if (!pend) base += 64;
                } else {
                    p = clq_pop(q, lane);
                    if (p < 0) break;
                }
                if (lane == 0) fo[p] = (uint8_t)(C_CHANGE | (phase == 0 ? C_SEED : 0));
                int nb = lane >> 2;
                int pn = nb == 0 ? p - ww : nb == 1 ? p - 1 : nb == 2 ? p + ww : p + 1;
                bool ok = false;
                float dist = 0.f;
                if (lane < 16) {
                    int pr = p / ww, pc = p - pr * ww;
                    int r = pr + (nb == 0 ? -1 : nb == 2 ? 1 : 0), cc = pc + (nb == 1 ? -1 : nb == 3 ? 1 : 0);
                    // window cells adjacent to a ring pixel always exist (margin range+1); frame-border cells are never INSIDE
                    ok = r >= 0 && r < wh && cc >= 0 && cc < ww && fo[pn] == C_INSIDE;
                }
                if (__ballot(ok)) {
                    if (ok) {
                        int qd = lane & 3;
                        int p1 = (qd & 1) ? pn + ww : pn - ww;
                        int p2 = (qd & 2) ? pn + 1 : pn - 1;
                        dist = cl_solve(fo, t, p1, p2);
                    }
                    float o = __shfl_xor(dist, 1, 64); dist = o < dist ? o : dist;
                    o = __shfl_xor(dist, 2, 64); dist = o < dist ? o : dist;
                    for (int k = 0; k < 4; k++) {
                        bool okk = __builtin_amdgcn_readlane((int)ok, k * 4) != 0;
                        if (!okk) continue;
                        float dk = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dist), k * 4));
                        int pk = __builtin_amdgcn_readlane(pn, k * 4);
                        if (lane == 0) { t[pk] = dk; fo[pk] = C_BAND; }
                        clq_push(q, dk, pk, lane);
                    }
                }
            }
        }
        for (int li = lane; li < cells; li += 64)
            if ((fo[li] & 0x7f) == C_CHANGE) t[li] = -t[li];

        // ---- pass 2: Telea march (icvTeleaInpaintFMM); seeds = band pixels in raster order, then the queue
        q.head = q.tail = 0;
        for (int phase = 0; phase < 2; phase++) {
            int base = 0;
            unsigned long long pend = 0;
            for (;;) {
                int p;
                if (phase == 0) {
                    while (!pend && base < cells) {
                        int li = base + lane;
                        pend = __ballot(li < cells && (fo[li] & C_SEED));
                        if (!pend) base += 64;
                    }
                    if (!pend) break;
                    int l = __ffsll((long long)pend) - 1;
                    pend &= pend - 1;
                    p = base + l;
                    if (!pend) base += 64;
                } else {
                    p = clq_pop(q, lane);
                    if (p < 0) break;
                    if (lane == 0) f[p] = C_KNOWN;
                }
                const int pr = p / ww, pc = p - pr * ww;
                for (int qn = 0; qn < 4; qn++) {
                    const int r = pr + (qn == 0 ? -1 : qn == 2 ? 1 : 0), cc = pc + (qn == 1 ? -1 : qn == 3 ? 1 : 0);
                    if (r < 0 || r >= wh || cc < 0 || cc >= ww) continue;
                    const int pi = r * ww + cc;
                    const int i = i0 + r, j = j0 + cc;      // padded frame coordinates
                    if (i <= 0 || j <= 0 || i >= er - 1 || j >= ec - 1) continue;
                    if (f[pi] != C_INSIDE) continue;
                    float dist;
                    {
                        int qd = lane & 3;
                        int p1 = (qd & 1) ? pi + ww : pi - ww;
                        int p2 = (qd & 2) ? pi + 1 : pi - 1;
                        float s = cl_solve(f, t, p1, p2);
                        float o = __shfl_xor(s, 1, 64); s = o < s ? o : s;
                        o = __shfl_xor(s, 2, 64); s = o < s ? o : s;
                        dist = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(s)));
                    }
                    if (lane == 0) t[pi] = dist;
                    const float tc = dist;
                    float gtx, gty;
                    {
                        bool kr = f[pi + 1] != C_INSIDE, kl = f[pi - 1] != C_INSIDE;
                        float tr = t[pi + 1], tl = t[pi - 1];
                        if (kr) gtx = kl ? __fmul_rn(__fsub_rn(tr, tl), 0.5f) : __fsub_rn(tr, tc);
                        else gtx = kl ? __fsub_rn(tc, tl) : 0.f;
                        bool kd = f[pi + ww] != C_INSIDE, ku = f[pi - ww] != C_INSIDE;
                        float td = t[pi + ww], tu = t[pi - ww];
                        if (kd) gty = ku ? __fmul_rn(__fsub_rn(td, tu), 0.5f) : __fsub_rn(td, tc);
                        else gty = ku ? __fsub_rn(tc, tu) : 0.f;
                    }
                    float sIa = 0, sJx = 0, sJy = 0, sS = 0;
                    for (int n0 = 0; n0 < nn; n0 += 64) {
                        int nidx = n0 + lane;
                        float cIa = 0.f, cJx = 0.f, cJy = 0.f, cS = 0.f;
                        if (nidx < nn) {
                            int dk = nidx / side - range, dl = nidx % side - range;
                            int k = i + dk, l = j + dl;                 // padded frame coordinates
                            if (k > 0 && l > 0 && k < er - 1 && l < ec - 1) {
                                int pk = (r + dk) * ww + (cc + dl);     // inside the window: margin range+1
                                if (f[pk] != C_INSIDE && (dl * dl + dk * dk <= r2)) {
                                    // OpenCV's index shifts at the first/last image row/column
                                    int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2);
                                    int lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                                    // image row/col (0-based) -> window cell: row + 1 - i0, col + 1 - j0
#define IMW(rr, c2) im[((rr) + 1 - i0) * ww + ((c2) + 1 - j0)]
                                    float ry = (float)(-dk), rx = (float)(-dl);
                                    float len2 = __fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry));
                                    float dstw = (float)(1. / (double)__fmul_rn(len2, __fsqrt_rn(len2)));
                                    float lev = (float)(1. / (1 + fabs((double)__fsub_rn(t[pk], tc))));
                                    float dir = __fadd_rn(__fmul_rn(rx, gtx), __fmul_rn(ry, gty));
                                    if (fabs((double)dir) <= 0.01) dir = 0.000001f;
                                    float wgt = fabsf(__fmul_rn(__fmul_rn(dstw, lev), dir));
                                    float gix, giy;
                                    bool kr = f[pk + 1] != C_INSIDE, kl = f[pk - 1] != C_INSIDE;
                                    if (kr) gix = kl ? __fmul_rn(__fsub_rn(IMW(km, lp + 1), IMW(km, lm - 1)), 2.0f) : __fsub_rn(IMW(km, lp + 1), IMW(km, lm));
                                    else gix = kl ? __fsub_rn(IMW(km, lp), IMW(km, lm - 1)) : 0.f;
                                    bool kd = f[pk + ww] != C_INSIDE, ku = f[pk - ww] != C_INSIDE;
                                    if (kd) giy = ku ? __fmul_rn(__fsub_rn(IMW(kp + 1, lm), IMW(km - 1, lm)), 2.0f) : __fsub_rn(IMW(kp + 1, lm), IMW(km, lm));
                                    else giy = ku ? __fsub_rn(IMW(kp, lm), IMW(km - 1, lm)) : 0.f;
                                    cIa = __fmul_rn(wgt, IMW(km, lm));
                                    cJx = __fmul_rn(wgt, __fmul_rn(gix, rx));
                                    cJy = __fmul_rn(wgt, __fmul_rn(giy, ry));
                                    cS = wgt;
#undef IMW
                                }
                            }
                        }
                        sIa += cl_dpp_sum_f32(cIa);
                        sJx -= cl_dpp_sum_f32(cJx);
                        sJy -= cl_dpp_sum_f32(cJy);
                        sS += cl_dpp_sum_f32(cS);
                    }
                    float Ia = sIa, Jx = sJx, Jy = sJy, s = sS + 1.0e-20f;
                    float val = (float)((double)__fdiv_rn(Ia, s) +
                                        (double)__fadd_rn(Jx, Jy) / (sqrt((double)__fadd_rn(__fmul_rn(Jx, Jx), __fmul_rn(Jy, Jy))) + (double)1.0e-20f));
                    if (lane == 0) { im[pi] = val; f[pi] = C_BAND; }
                    clq_push(q, dist, pi, lane);
                }
            }
        }
        // ---- write back this cluster's hole pixels
        for (int li = lane; li < cells; li += 64) {
            int r = li / ww, cc = li - r * ww;
            int gi = i0 + r, gj = j0 + cc;
            if (gi >= 1 && gi <= h && gj >= 1 && gj <= w) {
                size_t gp = (size_t)(gi - 1) * w + (gj - 1);
                if (bad[gp] && lab[gp] == rootp) img[gp] = im[li];
            }
        }
    }
}

// scratch layout (bytes per frame, see inpaint_cl_scratch_bytes_per_frame):
//   dil u8[P] | labels i32[P] | xmin,ymin,xmax,ymax i32[P] x4 | list i32[P] | big u8[P] | bad_big u8[P] | count,cursor
size_t inpaint_cl_scratch_bytes_per_frame(int h, int w)
{
    size_t P = (size_t)h * w;
    return P * (1 + 4 + 16 + 4 + 1 + 1) + 64;
}

void launch_inpaint_clusters(float *img, const uint8_t *bad, int range, void *scratch, uint8_t **bad_big_out, int B, int h, int w,
                             hipStream_t st)
{
    const int P = h * w;
    const size_t n = (size_t)B * P;
    uint8_t *base = (uint8_t *)scratch;
    uint8_t *dil = base; base += (n + 255) & ~(size_t)255;
    int32_t *labels = (int32_t *)base; base += n * 4;
    int32_t *xmin = (int32_t *)base; base += n * 4;
    int32_t *ymin = (int32_t *)base; base += n * 4;
    int32_t *xmax = (int32_t *)base; base += n * 4;
    int32_t *ymax = (int32_t *)base; base += n * 4;
    int32_t *list = (int32_t *)base; base += n * 4;
    uint8_t *big = base; base += (n + 255) & ~(size_t)255;
    uint8_t *bad_big = base; base += (n + 255) & ~(size_t)255;
    int32_t *count = (int32_t *)base; base += (size_t)B * 4;
    int32_t *cursor = (int32_t *)base;
    RowSpanSE se;
    int R = range + 1;   // hole pixels interact only within Chebyshev distance 2*range+1; R = range would already separate them
    se.k = 2 * R + 1;
    for (int i = 0; i < se.k; i++) { se.lo[i] = (int8_t)(-R); se.hi[i] = (int8_t)R; }
    launch_morph(bad, dil, B, h, w, se, true, nullptr, nullptr, st);
    launch_cc_label(dil, labels, B, h, w, st);
    hipMemsetAsync(xmin, 0x7f, n * 8, st);          // xmin, ymin
    hipMemsetAsync(xmax, 0, n * 8, st);             // xmax, ymax
    hipMemsetAsync(count, 0, (size_t)B * 8, st);    // count, cursor
    dim3 g((P + 255) / 256, B);
    hipLaunchKernelGGL(k_cluster_bbox, g, dim3(256), 0, st, bad, labels, xmin, ymin, xmax, ymax, h, w);
    hipLaunchKernelGGL(k_cluster_list, g, dim3(256), 0, st, labels, xmin, ymin, xmax, ymax, list, count, big, range, h, w);
    hipLaunchKernelGGL(k_split_bad, g, dim3(256), 0, st, bad, labels, big, bad_big, P);
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute((const void *)k_telea_clusters, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k_telea_clusters, dim3(4, B), dim3(64 * CL_WAVES), (size_t)CL_WMAX * 16, st, img, bad, labels, list, count, cursor, xmin,
                       ymin, xmax, ymax, range, h, w);
    *bad_big_out = bad_big;
}

bool inpaint_clusters_supported(int range) { return range >= 1 && range + 1 <= 16; }

}  // namespace vf
