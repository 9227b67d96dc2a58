// cv2.inpaint(img_f32, mask, radius, INPAINT_TELEA) on the GPU (shape_ftp.py:652-666, :1199).
//
// Telea's fast-marching inpaint is ordered by a stable priority queue keyed on (T, push sequence)
// [OpenCV photo/inpaint.cpp CvPriorityQueueFloat], so the march itself is sequential per frame.  One
// wavefront owns one frame: the queue lives in LDS and is popped by a 64-lane parallel arg-min, the
// four quadrant solves of a pixel run on four lanes, and the (2r+1)^2 neighbourhood sums of Telea's
// estimator run one neighbour per lane with wave reductions.  Frames are independent, so a batch
// fills the chip with one wave per frame.
//
// Mutable per-frame state: flag planes (LDS when (h+2)*(w+2) fits, else global), T field and image in
// global memory.  Global mutable words are read with agent-scope relaxed atomic loads (L2-served)
// and every store is drained (workgroup fence) before the next dependent read.
#include <cstdio>
#include "kernels.hpp"
#include "telea_common.hpp"

namespace vf {

constexpr int TQ_CAP = 6144;                 // LDS queue capacity (entries)
constexpr uint8_t T_KNOWN = 0, T_BAND = 1, T_INSIDE = 2, T_CHANGE = 3, T_SEED = 0x80;   // T_SEED: bit flag, initial band

__device__ inline float ldc(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
template <bool LF>
__device__ inline uint8_t ldf(const uint8_t *f, int i)
{
    if (LF) return f[i];
    return __hip_atomic_load(f + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ inline void drain() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// Stable priority queue kept as a SORTED array in LDS (the structure of OpenCV's CvPriorityQueueFloat): FMM
// pushes are nearly monotone in T, so an insertion shifts only the few trailing entries with a larger T
// (found and moved by the 64 lanes at once) and lands AFTER every entry with T' <= T (FIFO among ties);
// pop is the head.  [head, tail) slides up; it is moved back to 0 when the array end is reached.
// Diagnostic shader-clock stamps / pop counts of frame 0: only in builds with -DVISTAF_DEBUG (the stamps are process-wide globals that
// concurrent sessions would race on, and s_memtime + a global store per phase is not free on a lone latency-bound wave).
#ifdef VISTAF_DEBUG
__device__ unsigned long long g_telea_dbg[16];
#define TSTAMP(i) do { if (b == 0 && lane == 0) g_telea_dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

__device__ inline uint32_t ldq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

struct TQueue {
    uint32_t *T;       // float bits of T (>= 0), ascending in [head, tail)
    uint32_t *idx;
    int head, tail, cap;
    int overflow;
};

template <bool LQ>
__device__ inline void tq_push(TQueue &q, float Tf, int idx, int lane)
{
    if (!LQ) drain();
    if (q.tail >= q.cap) {
        if (q.head == 0) { q.overflow = 1; return; }
        int n = q.tail - q.head;
        for (int j0 = 0; j0 < n; j0 += 64) {
            int j = j0 + lane;
            uint32_t tv = 0, iv = 0;
            if (j < n) { tv = LQ ? q.T[j + q.head] : ldq(q.T + j + q.head); iv = LQ ? q.idx[j + q.head] : ldq(q.idx + j + q.head); }
            if (!LQ) drain();
            if (j < n) { q.T[j] = tv; q.idx[j] = iv; }
            if (!LQ) drain();
        }
        q.head = 0; q.tail = n;
    }
    const uint32_t tb = __float_as_uint(Tf);
    int k = 0;
    for (;;) {
        int j = q.tail - 1 - k - lane;
        bool in = j >= q.head;
        uint32_t tv = in ? (LQ ? q.T[j] : ldq(q.T + j)) : 0u;
        uint32_t iv = in ? (LQ ? q.idx[j] : ldq(q.idx + j)) : 0u;
        unsigned long long g = __ballot(in && tv > tb);
        int c = (g == ~0ull) ? 64 : (int)(__ffsll((long long)~g) - 1);     // leading run of "greater" entries
        if (!LQ) drain();
        if (lane < c) { q.T[j + 1] = tv; q.idx[j + 1] = iv; }
        if (!LQ) drain();
        k += c;
        if (c < 64) break;
    }
    if (lane == 0) { q.T[q.tail - k] = tb; q.idx[q.tail - k] = (uint32_t)idx; }
    q.tail++;
}

// pop the smallest (T, then oldest); -1 when empty (uniform)
template <bool LQ>
__device__ inline int tq_pop(TQueue &q, int lane)
{
    (void)lane;
    if (q.head == q.tail) return -1;
    if (!LQ) drain();
    int idx = (int)(LQ ? q.idx[q.head] : ldq(q.idx + q.head));
    q.head++;
    return __builtin_amdgcn_readfirstlane(idx);
}

// full-wave float sum (DPP butterfly; result uniform)
__device__ inline float dpp_sum_f32(float x)
{
    int v = __float_as_int(x);
#define VF_ADD(ctrl, rm)                                                                          \
    v = __float_as_int(__int_as_float(v) + __int_as_float(__builtin_amdgcn_update_dpp(0, v, ctrl, rm, 0xf, false)));
    VF_ADD(0xB1, 0xf) VF_ADD(0x4E, 0xf) VF_ADD(0x141, 0xf) VF_ADD(0x140, 0xf) VF_ADD(0x142, 0xa) VF_ADD(0x143, 0xc)
#undef VF_ADD
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}

template <bool LF>
__device__ inline float fmm_solve(const uint8_t *f, const float *t, int p1, int p2)
{
    double a11 = ldc(t + p1), a22 = ldc(t + p2);
    double m12 = a11 < a22 ? a11 : a22;
    bool k1 = ldf<LF>(f, p1) != T_INSIDE, k2 = ldf<LF>(f, p2) != T_INSIDE;
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

// dist of pixel p = min over the four quadrants; lanes 0..3 each solve one quadrant
template <bool LF>
__device__ inline float fmm_dist(const uint8_t *f, const float *t, int p, int ec, int lane)
{
    int q = lane & 3;
    int p1 = (q & 1) ? p + ec : p - ec;      // (i-1,j) (i+1,j) (i-1,j) (i+1,j)
    int p2 = (q & 2) ? p + 1 : p - 1;        // (i,j-1) (i,j-1) (i,j+1) (i,j+1)
    float s = fmm_solve<LF>(f, t, p1, p2);
    float o = __shfl_xor(s, 1, 64); s = o < s ? o : s;
    o = __shfl_xor(s, 2, 64); s = o < s ? o : s;
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(s)));
}

// Parallel preparation (one thread per padded cell): Telea flags f (INSIDE on the hole), outside-pass flags fo
// (ring = within Chebyshev `range` of the hole, seeds = 4-neighbour band), T = 1e6 / 0 on the band, hole count.
__global__ void k_telea_prep(const uint8_t *__restrict__ bad_all, uint8_t *__restrict__ gflags, float *__restrict__ gT,
                             int32_t *__restrict__ nbad_all, const int32_t *__restrict__ only, int range, int h, int w)
{
    const int er = h + 2, ec = w + 2, en = er * ec;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (i >= en) return;
    if (only && !only[b]) return;      // frame already inpainted by the window kernel: its hole count stays 0
    const uint8_t *bad = bad_all + b * (size_t)h * w;
    int y = i / ec, x = i - y * ec;
    auto hole = [&](int yy, int xx) -> bool { return yy >= 1 && yy <= h && xx >= 1 && xx <= w && bad[(size_t)(yy - 1) * w + (xx - 1)] != 0; };
    bool interior = (y >= 1 && y <= h && x >= 1 && x <= w);
    bool in = hole(y, x);
    uint8_t fv = in ? T_INSIDE : T_KNOWN, fov = T_KNOWN;
    float tv = 1.0e6f;
    if (interior && !in) {
        bool band = hole(y, x - 1) || hole(y, x + 1) || hole(y - 1, x) || hole(y + 1, x);
        if (band) { fov = T_KNOWN | T_SEED; tv = 0.f; }   // seeds: known, T = 0 (Heap->Add(band) / Out->Add(band))
        else {
            bool near = false;
            for (int a = -range; a <= range && !near; a++)
                for (int c = -range; c <= range; c++)
                    if (hole(y + a, x + c)) { near = true; break; }
            if (near) fov = T_INSIDE;
        }
    }
    gflags[b * (size_t)en * 2 + i] = fv;
    gflags[b * (size_t)en * 2 + en + i] = fov;
    gT[b * (size_t)en + i] = tv;
    if (in) atomicAdd(&nbad_all[b], 1);
}

// Window of a march in padded coordinates (inclusive) and the cluster it is restricted to: lab == nullptr marches every hole pixel of the
// frame (the window is then the whole padded frame); otherwise only cells whose label (component of the dilated hole mask, k_inpaint_cl.hip)
// equals `root` belong to this march -- other clusters may reach into the bounding window, their cells are simply skipped by the scans, and
// the march itself never leaves its own cluster (rings of different clusters are at least three cells apart).
struct TeleaScan { int i0, i1, j0, j1; const int32_t *lab; int root; };

template <bool LF>
__device__ __attribute__((always_inline)) inline void telea_march_global(float *img, int range, uint8_t *f, uint8_t *fo, float *t, TQueue &q,
                                                                         const TeleaScan sc, int h, int w, int lane, size_t b, bool round_u8 = false)
{
    const int er = h + 2, ec = w + 2;
    // raster scan of the window, 64 columns at a time: the cells of this march that satisfy `pred`
    auto mine = [&](int row, int col) -> bool { return sc.lab == nullptr || (row >= 1 && row <= h && col >= 1 && col <= w && sc.lab[(size_t)(row - 1) * w + (col - 1)] == sc.root); };
    TSTAMP(1);
    unsigned long long npop1 = 0, npop2 = 0;
    // ---- pass 1: outside T field.  Seeds (band, T = 0) pop first, in raster order.
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1) TSTAMP(2);
        int base = 0, row = sc.i0, cb = sc.j0;
        unsigned long long pend = 0;
        for (;;) {
            int p;
            if (phase == 0) {
                while (!pend && row <= sc.i1) {
                    const int col = cb + lane;
                    const int i = row * ec + col;
                    pend = __ballot(col <= sc.j1 && (ldf<LF>(fo, i) & T_SEED) && mine(row, col));
                    base = row * ec + cb;
                    cb += 64;
                    if (cb > sc.j1) { cb = sc.j0; row++; }
                }
                if (!pend) break;
                int l = __ffsll((long long)pend) - 1;
                pend &= pend - 1;
                p = base + l;
            } else {
                p = tq_pop<LF>(q, lane);
                if (p < 0) break;
            }
            npop1++;
            if (lane == 0) fo[p] = (uint8_t)(T_CHANGE | (phase == 0 ? T_SEED : 0));
            // the four 4-neighbours are independent of one another in this pass: lanes 0..15 = 4 pixels x 4 quadrants
            int nb = lane >> 2;
            int pn = nb == 0 ? p - ec : nb == 1 ? p - 1 : nb == 2 ? p + ec : p + 1;
            bool ok = false;
            float dist = 0.f;
            if (lane < 16) {
                int y = pn / ec, x = pn - y * ec;
                ok = (y > 0 && x > 0 && y < er - 1 && x < ec - 1) && ldf<LF>(fo, pn) == T_INSIDE;
            }
            if (__ballot(ok)) {
                drain();
                if (ok) {
                    int qd = lane & 3;
                    int p1 = (qd & 1) ? pn + ec : pn - ec;
                    int p2 = (qd & 2) ? pn + 1 : pn - 1;
                    dist = fmm_solve<LF>(fo, t, p1, p2);
                }
                float o = __shfl_xor(dist, 1, 64); dist = o < dist ? o : dist;
                o = __shfl_xor(dist, 2, 64); dist = o < dist ? o : dist;
                for (int k = 0; k < 4; k++) {
                    bool okk = __builtin_amdgcn_readlane((int)ok, k * 4) != 0;
                    if (!okk) continue;
                    float dk = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dist), k * 4));
                    int pk = __builtin_amdgcn_readlane(pn, k * 4);
                    if (lane == 0) { t[pk] = dk; fo[pk] = T_BAND; }
                    tq_push<LF>(q, dk, pk, lane);
                }
            }
            if (LF) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
    }
    drain();
    __syncthreads();
    TSTAMP(3);
    // negate T where the outside pass ran (CHANGE), seeds keep T = 0
    for (int row = sc.i0; row <= sc.i1; row++)
        for (int col = sc.j0 + lane; col <= sc.j1; col += 64) {
            const int i = row * ec + col;
            if ((ldf<LF>(fo, i) & 0x7f) == T_CHANGE && mine(row, col)) { float v = ldc(t + i); t[i] = -v; }
        }
    drain();
    __syncthreads();

    TSTAMP(4);
    // ---- pass 2: Telea march.  Seeds = band pixels (raster order), then the queue.
    q.head = q.tail = 0;
    const int r2 = range * range;
    // 1 / |r|^3 of this lane's neighbour offset (first two 64-neighbour chunks), hoisted out of the march
    float pre_dstw[2];
    for (int c2 = 0; c2 < 2; c2++) {
        int nidx = c2 * 64 + lane, sd = 2 * range + 1;
        float ry = (float)(range - nidx / sd), rx = (float)(range - nidx % sd);
        float len2 = __fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry));
        pre_dstw[c2] = len2 > 0.f ? (float)(1. / (double)__fmul_rn(len2, sqrtf(len2))) : 0.f;
    }
    const int side = 2 * range + 1, nn = side * side;
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1) TSTAMP(5);
        int base = 0, row = sc.i0, cb = sc.j0;
        unsigned long long pend = 0;
        for (;;) {
            int p;
            if (phase == 0) {
                while (!pend && row <= sc.i1) {
                    const int col = cb + lane;
                    const int i = row * ec + col;
                    pend = __ballot(col <= sc.j1 && (ldf<LF>(fo, i) & T_SEED) && mine(row, col));
                    base = row * ec + cb;
                    cb += 64;
                    if (cb > sc.j1) { cb = sc.j0; row++; }
                }
                if (!pend) break;
                int l = __ffsll((long long)pend) - 1;
                pend &= pend - 1;
                p = base + l;
            } else {
                p = tq_pop<LF>(q, lane);
                if (p < 0) break;
            }
            npop2++;
            if (phase == 1 && lane == 0) f[p] = T_KNOWN;
            if (LF) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else drain();
            for (int qn = 0; qn < 4; qn++) {
                int pi = qn == 0 ? p - ec : qn == 1 ? p - 1 : qn == 2 ? p + ec : p + 1;
                int i = pi / ec, j = pi - i * ec;
                if (i <= 0 || j <= 0 || i >= er - 1 || j >= ec - 1) continue;
                if (ldf<LF>(f, pi) != T_INSIDE) continue;
                drain();
                float dist = fmm_dist<LF>(f, t, pi, ec, lane);
                if (lane == 0) t[pi] = dist;
                // gradT at (i,j): uses the T just computed for the centre
                float tc = dist;
                float gtx, gty;
                {
                    bool kr = ldf<LF>(f, pi + 1) != T_INSIDE, kl = ldf<LF>(f, pi - 1) != T_INSIDE;
                    float tr = ldc(t + pi + 1), tl = ldc(t + pi - 1);
                    if (kr) gtx = kl ? __fmul_rn(__fsub_rn(tr, tl), 0.5f) : __fsub_rn(tr, tc);
                    else gtx = kl ? __fsub_rn(tc, tl) : 0.f;
                    bool kd = ldf<LF>(f, pi + ec) != T_INSIDE, ku = ldf<LF>(f, pi - ec) != T_INSIDE;
                    float td = ldc(t + pi + ec), tu = ldc(t + pi - ec);
                    if (kd) gty = ku ? __fmul_rn(__fsub_rn(td, tu), 0.5f) : __fsub_rn(td, tc);
                    else gty = ku ? __fsub_rn(tc, tu) : 0.f;
                }
                float sIa = 0, sJx = 0, sJy = 0, sS = 1.0e-20f;     // running sums in OpenCV's order (wn_seq_sum_n): chunk after chunk, lane after lane
                for (int n0 = 0; n0 < nn; n0 += 64) {
                    int nidx = n0 + lane;
                    float cIa = 0.f, cJx = 0.f, cJy = 0.f, cS = 0.f;
                    if (nidx < nn) {
                        int k = i - range + nidx / side, l = j - range + nidx % side;
                        if (k > 0 && l > 0 && k < er - 1 && l < ec - 1) {
                            int pk = k * ec + l;
                            if (ldf<LF>(f, pk) != T_INSIDE && ((l - j) * (l - j) + (k - i) * (k - i) <= r2)) {
                                int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2);
                                int lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                                float ry = (float)(i - k), rx = (float)(j - l);
                                float len2 = __fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry));
                                float dstw = n0 == 0 ? pre_dstw[0] : (n0 == 64 ? pre_dstw[1] : (float)(1. / (double)__fmul_rn(len2, sqrtf(len2))));
                                float tk = (pk == pi) ? tc : ldc(t + pk);
                                float lev = (float)(1. / (1 + fabs((double)__fsub_rn(tk, tc))));
                                float dir = __fadd_rn(__fmul_rn(rx, gtx), __fmul_rn(ry, gty));
                                if (fabsf(dir) <= 0.01f) dir = 0.000001f;   // float(0.01) < 0.01: same set of floats as the double compare
                                float wgt = fabsf(__fmul_rn(__fmul_rn(dstw, lev), dir));
                                float gix, giy;
                                bool kr = ldf<LF>(f, pk + 1) != T_INSIDE, kl = ldf<LF>(f, pk - 1) != T_INSIDE;
                                if (kr) gix = kl ? __fmul_rn(__fsub_rn(ldc(img + (size_t)km * w + lp + 1), ldc(img + (size_t)km * w + lm - 1)), 2.0f)
                                                 : __fsub_rn(ldc(img + (size_t)km * w + lp + 1), ldc(img + (size_t)km * w + lm));
                                else gix = kl ? __fsub_rn(ldc(img + (size_t)km * w + lp), ldc(img + (size_t)km * w + lm - 1)) : 0.f;
                                bool kd = ldf<LF>(f, pk + ec) != T_INSIDE, ku = ldf<LF>(f, pk - ec) != T_INSIDE;
                                if (kd) giy = ku ? __fmul_rn(__fsub_rn(ldc(img + (size_t)(kp + 1) * w + lm), ldc(img + (size_t)(km - 1) * w + lm)), 2.0f)
                                                 : __fsub_rn(ldc(img + (size_t)(kp + 1) * w + lm), ldc(img + (size_t)km * w + lm));
                                else giy = ku ? __fsub_rn(ldc(img + (size_t)kp * w + lm), ldc(img + (size_t)(km - 1) * w + lm)) : 0.f;
                                cIa = __fmul_rn(wgt, ldc(img + (size_t)km * w + lm));
                                cJx = __fmul_rn(wgt, __fmul_rn(gix, rx));
                                cJy = __fmul_rn(wgt, __fmul_rn(giy, ry));
                                cS = wgt;
                            }
                        }
                    }
                    const int nl = nn - n0 < 64 ? nn - n0 : 64;
                    sIa = wn_seq_sum_n(cIa, sIa, nl, lane);
                    sJx = wn_seq_sum_n(-cJx, sJx, nl, lane);
                    sJy = wn_seq_sum_n(-cJy, sJy, nl, lane);
                    sS = wn_seq_sum_n(cS, sS, nl, lane);
                }
                const float val = round_u8 ? telea_estimate_u8(sIa, sJx, sJy, sS) : telea_estimate(sIa, sJx, sJy, sS);
                if (lane == 0) { img[(size_t)(i - 1) * w + (j - 1)] = val; f[pi] = T_BAND; }
                tq_push<LF>(q, dist, pi, lane);
                if (LF) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
        }
    }
    TSTAMP(6);
#ifdef VISTAF_DEBUG
    if (b == 0 && lane == 0) { g_telea_dbg[8] = npop1; g_telea_dbg[9] = npop2; }
#endif
    (void)npop1; (void)npop2;
}

template <bool LF>
__global__ __launch_bounds__(64) void k_telea(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all, int range,
                                              uint8_t *gflags, float *gT, uint32_t *gqueue, const int32_t *__restrict__ nbad_all, int32_t *status, int h,
                                              int w, int round_u8)
{
    extern __shared__ unsigned char lds_raw[];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    const int er = h + 2, ec = w + 2, en = er * ec, P = h * w;
    float *img = img_all + b * (size_t)P;
    float *t = gT + b * (size_t)en;
    TQueue q;
    if (LF) { q.T = (uint32_t *)lds_raw; q.idx = (uint32_t *)(lds_raw + (size_t)TQ_CAP * 4); q.cap = TQ_CAP; }
    else { q.T = gqueue + b * (size_t)en * 2; q.idx = q.T + en; q.cap = en; }   // large frames: queue in global memory
    q.head = q.tail = 0; q.overflow = 0;
    uint8_t *f, *fo;
    if (LF) { f = lds_raw + (size_t)TQ_CAP * 8; fo = f + ((en + 15) & ~15); }
    else { f = gflags + b * (size_t)en * 2; fo = f + en; }
    TSTAMP(0);
    // ---- flags and the initial T field were prepared by k_telea_prep (all CUs); stage the flag planes into LDS
    if (nbad_all[b] == 0) return;
    if (LF) {
        const uint8_t *gsrc = gflags + b * (size_t)en * 2;
        for (int i = lane; i < en; i += 64) { f[i] = gsrc[i]; fo[i] = gsrc[en + i]; }
        __syncthreads();
    }
    const TeleaScan sc = {0, er - 1, 0, ec - 1, nullptr, 0};
    telea_march_global<LF>(img, range, f, fo, t, q, sc, h, w, lane, b, round_u8 != 0);
    if (q.overflow && lane == 0) status[b] = 2;
    (void)bad_all;
}

#ifdef VISTAF_DEBUG
void telea_debug_dump()
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_telea_dbg), sizeof(h)) != hipSuccess) return;
    printf("[telea dbg] cycles: init %llu | pass1 seeds %llu | pass1 queue %llu | negate %llu | pass2 seeds %llu | pass2 queue %llu | pops %llu / %llu | bad %llu\n",
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[8], h[9], h[10]);
}
#endif

size_t inpaint_scratch_bytes_per_frame(int h, int w)
{
    size_t en = (size_t)(h + 2) * (w + 2);
    // the whole-frame kernel's planes, or those of the big-cluster kernel (k_inpaint_big.hip): never both in one step
    return std::max(en * sizeof(float) + 2 * en + 8 * en /*global queue (large frames)*/ + 64 + 8 + 256, inpaint_big_scratch_bytes_per_frame(h, w));
}

static size_t telea_lds_bytes(int h, int w)
{
    size_t en = (size_t)(h + 2) * (w + 2);
    return (size_t)TQ_CAP * 8 + 2 * ((en + 15) & ~(size_t)15);
}

void launch_inpaint_telea(float *img, const uint8_t *bad, int range, void *scratch, int32_t *status, const int32_t *only, int B, int h,
                          int w, hipStream_t st, bool round_u8)
{
    size_t en = (size_t)(h + 2) * (w + 2);
    // scratch layout: [B*en floats T][B*2*en bytes flags]
    float *gT = (float *)scratch;
    uint8_t *gflags = (uint8_t *)scratch + (size_t)B * en * sizeof(float);
    uint32_t *gqueue = (uint32_t *)((((uintptr_t)scratch + (size_t)B * en * sizeof(float) + (size_t)B * en * 2) + 255) & ~(uintptr_t)255);
    int32_t *nbad = (int32_t *)(gqueue + (size_t)B * en * 2);
    hipMemsetAsync(nbad, 0, sizeof(int32_t) * B, st);
    hipLaunchKernelGGL(k_telea_prep, dim3((unsigned)((en + 255) / 256), B), dim3(256), 0, st, bad, gflags, gT, nbad, only, range, h, w);
    size_t lds_full = telea_lds_bytes(h, w);
    if (lds_full <= 160 * 1024) {
        static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_telea<true>, 160 * 1024);
        hipLaunchKernelGGL(k_telea<true>, dim3(B), dim3(64), lds_full, st, img, bad, range, gflags, gT, gqueue, nbad, status, h, w, round_u8 ? 1 : 0);
    } else {
        hipLaunchKernelGGL(k_telea<false>, dim3(B), dim3(64), 0, st, img, bad, range, gflags, gT, gqueue, nbad, status, h, w, round_u8 ? 1 : 0);
    }
}

}  // namespace vf
