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
#include <type_traits>
#include "telea_common.hpp"

namespace vf {

constexpr int WN_LOAD_U = 8;         // window cells per lane in flight while the window is loaded
constexpr int WN_RING_U = 6;         // inpaint ranges up to this have their ring taps unrolled (independent loads)
constexpr int WN_QCAP = 4096;       // live queue entries (8 B each)
constexpr int WN_CELLS = 14464;     // window cells (9 B each: T f32, image f32, flags u8)
// First tier: 110.75 KB.  It leaves 49 KB of the CU's LDS to the kernels of the other sessions in flight (exact selection 37 KB, column
// polyfit 44 KB, fused blur 26 KB ...), so a CU that marches a frame keeps working on another batch's stages instead of idling 3 SIMDs.
// Frames whose window or queue does not fit are flagged and marched again by the full-size tier on a small grid.
constexpr int WN1_CELLS = 10752;
constexpr int WN1_QCAP = 2048;
constexpr int WN2_GRID = 32;

// diagnostic shader-clock stamps per frame: only in builds with -DVISTAF_DEBUG (see k_inpaint.hip)
#ifdef VISTAF_DEBUG
__device__ unsigned long long g_win_dbg[1024][16];
#define WSTAMP(i) do { if (!CL && lane == 0 && b < 1024) g_win_dbg[b][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif


// box planes [4][B]: xmin, ymin (start 0x7f7f7f7f), xmax, ymax (start 0) of the hole pixels of each frame
// VEC: 16 mask bytes per thread (P must be a multiple of 16); most words are zero, coordinates are only formed for set bytes
template <bool VEC>
__global__ __launch_bounds__(256) void k_bad_bbox(const uint8_t *__restrict__ bad, int32_t *__restrict__ box, int B, int h, int w)
{
    const int b = blockIdx.y;
    const int P = h * w;
    int x0 = 0x7f7f7f7f, y0 = 0x7f7f7f7f, x1 = 0, y1 = 0;
    bool any = false;
    if (VEC) {
        const int q = blockIdx.x * blockDim.x + threadIdx.x;              // 16-byte word of the frame
        if (q * 16 < P) {
            const uint4 v = ((const uint4 *)(bad + (size_t)b * P))[q];
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
            if (v.x | v.y | v.z | v.w) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu) {
                        const int p = q * 16 + k, y = p / w, x = p - y * w;
                        x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
                        any = true;
                    }
                }
            }
        }
    } else {
        const int p = blockIdx.x * blockDim.x + threadIdx.x;
        if (p < P && bad[(size_t)b * P + p]) { y0 = y1 = p / w; x0 = x1 = p - y0 * w; any = true; }
    }
    if (!__ballot(any)) return;
    for (int o = 32; o; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&box[b], x0); atomicMin(&box[B + b], y0); atomicMax(&box[2 * B + b], x1); atomicMax(&box[3 * B + b], y1);
    }
}

// (nothing has been written back).
// always_inline: called from three kernels; left as a call the march's one-basic-block fill is laid out differently and the window
// kernel gets 15 % slower (3.91 against 3.40 ms per batch, measured)
template <bool CL>
__device__ __attribute__((always_inline)) inline bool wn_march(float *__restrict__ img, const uint8_t *__restrict__ bad, const int32_t *__restrict__ lab, int rootp,
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

    // ---- load window: hole / border bits, T = 1e6, image.  Cells in flat order, WN_LOAD_U per lane in flight (a row-by-row loop is one
    // memory round trip per 64 cells on this lone wave)
    {
        const uint32_t mg_ww = (uint32_t)(0x100000000ull / (unsigned)ww) + 1u;       // li / ww == umulhi(li, mg_ww) for li < 2^16
        for (int base = 0; base < cells; base += 64 * WN_LOAD_U) {
            float v[WN_LOAD_U];
            uint8_t bd[WN_LOAD_U];
            int lab_v[WN_LOAD_U];
            bool interior[WN_LOAD_U];
#pragma unroll
            for (int k = 0; k < WN_LOAD_U; k++) {
                const int li = base + k * 64 + lane;
                const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
                const int gi = i0 + r, gj = j0 + cc;
                interior[k] = li < cells && gi >= 1 && gi <= h && gj >= 1 && gj <= w;
                const size_t gp = interior[k] ? (size_t)(gi - 1) * w + (gj - 1) : 0;
                v[k] = img[gp];
                bd[k] = bad[gp];
                lab_v[k] = CL ? lab[gp] : 0;
            }
#pragma unroll
            for (int k = 0; k < WN_LOAD_U; k++) {
                const int li = base + k * 64 + lane;
                if (li >= cells) continue;
                uint8_t b8 = bd[k];
                if (CL) b8 = (b8 && lab_v[k] == rootp) ? 1 : 0;
                im[li] = interior[k] ? v[k] : 0.f;
                f[li] = !interior[k] ? W_BORDER : b8 ? W_HOLE : (uint8_t)0;
                t[li] = 1.0e6f;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    WSTAMP(1);
    // ---- ring = within Chebyshev `range` of the hole (separable: rows, then columns).  Cells in flat order, two per lane and iteration;
    // every tap of a cell is an independent load (clamped index + select: a loop over [-range, range] or a short-circuit test would be one
    // dependent LDS round trip per tap on this lone wave).
    {
        const uint32_t mg_ww = (uint32_t)(0x100000000ull / (unsigned)ww) + 1u;
        const bool unrolled = range <= WN_RING_U;
        for (int base = 0; base < cells; base += 128) {
            uint8_t any[2], me[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int li = min(base + k * 64 + lane, cells - 1);
                const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
                me[k] = f[li];
                uint8_t a = 0;
                if (unrolled) {
#pragma unroll
                    for (int d = -WN_RING_U; d <= WN_RING_U; d++) {
                        const int c2 = cc + d;
                        const bool in = d >= -range && d <= range && c2 >= 0 && c2 < ww;
                        const uint8_t v = f[in ? li + d : li];
                        a |= in ? v : (uint8_t)0;
                    }
                } else {
                    const int lo = max(0, cc - range), hi = min(ww - 1, cc + range);
                    for (int c2 = lo; c2 <= hi; c2++) a |= f[r * ww + c2];
                }
                any[k] = a;
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int li = base + k * 64 + lane;
                if (li < cells && (any[k] & W_HOLE)) f[li] = me[k] | W_ROW;      // neighbours only look at bit 6, which this never changes
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < cells; base += 128) {
            uint8_t any[2], me[2], nb4[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int li = min(base + k * 64 + lane, cells - 1);
                const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
                me[k] = f[li];
                // window edge cells are at distance range+1 from the hole: never band, their neighbours are not needed (a clamped index
                // reads the cell itself, which only matters when it is not a hole pixel)
                nb4[k] = f[cc > 0 ? li - 1 : li] | f[cc < ww - 1 ? li + 1 : li] | f[r > 0 ? li - ww : li] | f[r < wh - 1 ? li + ww : li];
                uint8_t a = 0;
                if (unrolled) {
#pragma unroll
                    for (int d = -WN_RING_U; d <= WN_RING_U; d++) {
                        const int r2 = r + d;
                        const bool in = d >= -range && d <= range && r2 >= 0 && r2 < wh;
                        const uint8_t v = f[in ? li + d * ww : li];
                        a |= in ? v : (uint8_t)0;
                    }
                } else {
                    const int lo = max(0, r - range), hi = min(wh - 1, r + range);
                    for (int r2 = lo; r2 <= hi; r2++) a |= f[r2 * ww + cc];
                }
                any[k] = a;
            }
            // reads of this iteration see W_ROW / W_HOLE only (bits 5 and 6), which the writes below never change
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int li = base + k * 64 + lane;
                if (li >= cells || (me[k] & (W_BORDER | W_HOLE))) continue;     // hole pixels are KNOWN for the outside pass
                if (nb4[k] & W_HOLE) { f[li] = me[k] | W_SEED; t[li] = 0.f; }
                else if (any[k] & W_ROW) f[li] = me[k] | W_INSIDE;               // writes bits 0-1 only
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    WSTAMP(2);
    TeleaWin win;
    win.t = t; win.im = im; win.f = f; win.ww = ww;
    WQ q;
    q.e = qe; q.hotL = hotL; q.ovf = 0; q.cap = qcap;
    wq_init(q);
    unsigned long long np1 = 0, np2 = 0, nfill = 0, ns1 = 0;
    // ---- pass 1: outside T field (icvCalcFMM, negate); seeds pop first in raster order, then the queue.
    // States here are those of OpenCV's `out` mask: ring = INSIDE, hole and everything else KNOWN.
    {
        FmmFlagState fst{t, f};
        WSTAMP(3);
        telea_fmm_pass(fst, q, f, cells, ww, lane, np1, ns1);
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
    const TeleaMarchConsts mc = telea_march_consts(lane, ww, range);
    WSTAMP(5);
    auto march = [&](auto small) {
        for (int phase = 0; phase < 2 && !q.ovf; phase++) {
            if (phase == 1) WSTAMP(6);
            int base = 0;
            unsigned long long pend = 0;
            for (;;) {
                int p = phase == 0 ? wn_next_seed(f, cells, base, pend, lane) : wq_pop(q);
                if (p < 0) break;
                np2++;
                nfill += telea_pop_march<decltype(small)::value>(win, mc, p, phase == 1, lane, [&](float T_, int idx_) { wq_push(q, T_, idx_, lane); });
            }
        }
    };
    // discs of the inpaint ranges 1..4 (5, 13, 29, 49 positions): straight-line estimator with that many lanes; anything else: chunk loop
    switch (mc.ndisc) {
    case 5: march(std::integral_constant<int, 5>{}); break;
    case 13: march(std::integral_constant<int, 13>{}); break;
    case 29: march(std::integral_constant<int, 29>{}); break;
    case 49: march(std::integral_constant<int, 49>{}); break;
    default: march(std::integral_constant<int, 0>{}); break;
    }
    WSTAMP(7);
#ifdef VISTAF_DEBUG
    if (!CL && lane == 0 && b < 1024) { g_win_dbg[b][8] = (np1 << 32) | np2; g_win_dbg[b][9] = (nfill << 32) | (unsigned)cells; g_win_dbg[b][10] = ns1; }
#endif
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

// Second tier: a small grid walks the batch and marches only the frames the first tier flagged (fb[b] == 1), clearing the flag on success.
template <bool RETRY>
__device__ __attribute__((always_inline)) inline void telea_window_body(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all,
                                                                        const int32_t *__restrict__ box, int32_t *__restrict__ fb, int range, int B,
                                                                        int h, int w, int cells_cap, int qcap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wn_lds[];
    const int lane = threadIdx.x;
    const int P = h * w;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        if (RETRY && fb[b] != 1) continue;
        const int xmin = box[b], ymin = box[B + b], xmax = box[2 * B + b], ymax = box[3 * B + b];
        if (xmin == 0x7f7f7f7f) continue;                        // no hole pixel: nothing to inpaint
        const int M = range + 1;
        // window in padded frame coordinates [i0, i1] x [j0, j1], not clipped (cells beyond the image: BORDER)
        const int i0 = ymin + 1 - M, i1 = ymax + 1 + M;
        const int j0 = xmin + 1 - M, j1 = xmax + 1 + M;
        const int wh = i1 - i0 + 1, ww = j1 - j0 + 1;
        bool ok = wh * ww <= cells_cap;
        if (ok) ok = wn_march<false>(img_all + (size_t)b * P, bad_all + (size_t)b * P, nullptr, 0, i0, j0, wh, ww, range, h, w, wn_lds, cells_cap, qcap, lane, b);
        if (lane == 0) fb[b] = ok ? 0 : 1;
    }
}
__global__ __launch_bounds__(64) void k_telea_window(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all, const int32_t *__restrict__ box,
                                                     int32_t *__restrict__ fb, int range, int B, int h, int w, int cells_cap, int qcap)
{
    telea_window_body<false>(img_all, bad_all, box, fb, range, B, h, w, cells_cap, qcap);
}
// its own symbol so that profiles list the (normally empty) second tier apart from the march proper
__global__ __launch_bounds__(64) void k_telea_window_retry(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all,
                                                           const int32_t *__restrict__ box, int32_t *__restrict__ fb, int range, int B, int h, int w,
                                                           int cells_cap, int qcap)
{
    telea_window_body<true>(img_all, bad_all, box, fb, range, B, h, w, cells_cap, qcap);
}

// Cluster front end: hole pixels farther apart than 2 * range + 3 never interact (rings and neighbourhoods reach
// range + 1), so every connected component of the hole mask dilated by a (2 * (range + 1) + 1)^2 square -- a CLUSTER --
// can be marched on its own small window with its own queue, in exactly the order the whole-frame queue would
// pop its pixels.  One wave per cluster, several clusters per CU: the batch turns from B sequential marches into
// thousands of short ones.  Clusters whose window or queue does not fit are flagged in big[] and left to
// k_telea_window, which then sees only their pixels.
constexpr int CL2_CELLS = 3072;     // window cells per cluster
constexpr int CL2_QCAP = 512;       // queue entries per cluster
constexpr int CL2_SLOTS = 48;       // workgroups per frame (each walks clusters c, c + slots, ...); more for small batches, see the launcher

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
    // a native crop has ~500 clusters: with few frames in the batch, more waves per frame (about five 32 KB windows fit a CU)
    const int slots = std::max(CL2_SLOTS, std::min(512, 4096 / std::max(B, 1)));
    hipLaunchKernelGGL(k_telea_clusters2, dim3(slots, B), dim3(64), lds, st, img, bad, labels, list, count, xmin, ymin, xmax, ymax, big, range, h, w);
}

#ifdef VISTAF_DEBUG
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
    {
        unsigned long long fd[8];
        if (hipMemcpyFromSymbol(fd, HIP_SYMBOL(g_fill_dbg), sizeof(fd)) == hipSuccess)
            printf("[telea fill dbg] frame 0 cycle sums over all calls so far: reads+solve %llu | terms %llu | ordered sums %llu | estimate %llu | push %llu\n", fd[0], fd[1], fd[2], fd[3], fd[4]);
    }
    for (int b : {0, worst}) {
        unsigned long long *x = hbuf[b];
        printf("[telea window dbg] frame %d cycles: load %llu | ring %llu | p1 seeds %llu | p1 queue %llu | negate %llu | p2 seeds %llu | p2 queue %llu | "
               "pops %llu (in %llu steps) / %llu | filled %llu | cells %llu | mean total over frames %.0f\n", b, x[1] - x[0], x[2] - x[1], x[3] - x[2], x[4] - x[3], x[5] - x[4], x[6] - x[5], x[7] - x[6],
               x[8] >> 32, x[10], x[8] & 0xffffffffull, x[9] >> 32, x[9] & 0xffffffffull, mean);
    }
}
#endif

size_t inpaint_win_scratch_bytes(int B) { return (size_t)B * 5 * sizeof(int32_t) + 256; }

// box scratch: [4][B] bbox planes + [B] fallback flags.  Returns the device pointer of the fallback flags.
int32_t *launch_inpaint_window(float *img, const uint8_t *bad, int range, void *scratch, int B, int h, int w, hipStream_t st, hipEvent_t ev_march,
                               bool two_tier, bool mw)
{
    int32_t *box = (int32_t *)scratch, *fb = box + 4 * (size_t)B;
    (void)hipMemsetAsync(box, 0x7f, (size_t)B * 8, st);
    (void)hipMemsetAsync(box + 2 * (size_t)B, 0, (size_t)B * 12, st);
    const int P = h * w;
    if (P % 16 == 0) hipLaunchKernelGGL(k_bad_bbox<true>, dim3((P / 16 + 255) / 256, B), dim3(256), 0, st, bad, box, B, h, w);
    else hipLaunchKernelGGL(k_bad_bbox<false>, dim3((P + 255) / 256, B), dim3(256), 0, st, bad, box, B, h, w);
    auto lds_bytes = [](int cells, int q) { return (size_t)cells * 9 + (size_t)q * 8 + 256; };
    static DynLdsOnce lds_once, lds_once_retry;
    ensure_dyn_lds(lds_once, (const void *)k_telea_window, 160 * 1024);
    ensure_dyn_lds(lds_once_retry, (const void *)k_telea_window_retry, 160 * 1024);
    if (ev_march) (void)hipEventRecord(ev_march, st);     // stage timing: the march starts here (the bbox pass belongs to the mask stage)
    if (mw && inpaint_window_mw_supported(range)) {
        // first tier: one 16-wave workgroup per frame (k_inpaint_mw.hip); what it hands back goes to the full-size single-wave march
        launch_telea_window_mw(img, bad, box, fb, range, B, h, w, st);
        hipLaunchKernelGGL(k_telea_window_retry, dim3(std::min(B, WN2_GRID)), dim3(64), lds_bytes(WN_CELLS, WN_QCAP), st, img, bad, box, fb, range, B, h, w,
                           WN_CELLS, WN_QCAP);
    } else if (two_tier) {
        hipLaunchKernelGGL(k_telea_window, dim3(B), dim3(64), lds_bytes(WN1_CELLS, WN1_QCAP), st, img, bad, box, fb, range, B, h, w, WN1_CELLS, WN1_QCAP);
        hipLaunchKernelGGL(k_telea_window_retry, dim3(std::min(B, WN2_GRID)), dim3(64), lds_bytes(WN_CELLS, WN_QCAP), st, img, bad, box, fb, range, B, h, w,
                           WN_CELLS, WN_QCAP);
    } else
        hipLaunchKernelGGL(k_telea_window, dim3(B), dim3(64), lds_bytes(WN_CELLS, WN_QCAP), st, img, bad, box, fb, range, B, h, w, WN_CELLS, WN_QCAP);
    return fb;
}

}  // namespace vf
