// cv2.inpaint(INPAINT_TELEA) on the frame's hole window -- one WORKGROUP of 16 waves per frame (shape_ftp.py:652-666).
//
// The single-wave window kernel (k_inpaint_win.hip) replays cv::inpaint's march pop by pop: every fill waits for the previous pixel's
// estimator although the estimator never decides what pops next.  What the march does in which order depends on the mask only
// (FastMarching_solve reads the flags and T, never the image), so this kernel splits it:
//   phase 0  all waves   load the window, build band / outside ring (the same cells, states and raster order as the single-wave kernel);
//   phase 1  wave 0      outside T field (icvCalcFMM with `negate`), states in the flag bytes;
//            wave 1      the march over the hole WITHOUT estimator (FmmOrderState): T of every hole pixel and the fill sequence.
//                        The two passes touch disjoint cells (ring cells have no hole neighbour, hole pixels no ring neighbour; the band
//                        pixels between them keep T = 0 and their seed bit), so they run side by side;
//   phase 2  all waves   negate the ring's T, switch the flag bytes to the march's states, count for every fill the earlier fills within
//                        Chebyshev distance range + 1 (the reach of a fill's reads);
//   phase 3  all waves   the estimates as a dataflow: a fill runs once its counter is 0, then decrements the counters of the later fills in
//                        reach and queues those that drop to 0.  Any two fills in reach of each other therefore run in march order and see
//                        each other's stores, fills farther apart commute -- the plane is the sequential march's bit for bit (each fill is
//                        telea_fill_known_T: the single-wave fill block minus solve and push);
//   phase 4  all waves   write the hole pixels back.
// The bench frames have ~1 400 fills per frame in ~200 dependence levels (tools/telea_dag.py), up to 17 fills wide.
// Frames whose window, queues or fill list do not fit are flagged in fb[] for the full-size single-wave tier (k_telea_window_retry).
#include <cstdio>
#include "kernels.hpp"
#include <type_traits>
#include "telea_common.hpp"

namespace vf {

constexpr int MW_WAVES = 16;
constexpr int MW_T = MW_WAVES * 64;
constexpr int MW_CELLS = 10752;     // window cells: 11 B each (T f32, image f32, fill number u16, flags u8)
constexpr int MW_QCAP = 2048;       // live queue entries of each of the two FMM passes
constexpr int MW_FILLS = 4096;      // hole pixels per frame
constexpr int MW_RING_U = 6;
enum { MWC_FAIL = 0, MWC_NFILL = 1, MWC_HEAD = 2, MWC_TAIL = 3, MWC_N = 16 };
constexpr size_t MW_LDS = (size_t)MW_QCAP * 16 + (size_t)MW_CELLS * 11 + (size_t)MW_FILLS * 2 + 2 * 64 * 4 + MWC_N * 4;
static_assert(MW_LDS <= 160 * 1024, "one CU's LDS");
static_assert(MW_FILLS * 4 <= MW_QCAP * 8 && MW_FILLS * 2 <= MW_QCAP * 8, "dependence counters / ready queue reuse the FMM queues");
static_assert(MW_FILLS < FI_NOHOLE, "fill numbers are 16-bit");

#ifdef VISTAF_DEBUG
__device__ unsigned long long g_mw_dbg[1024][16];
#define MSTAMP(i) do { if (threadIdx.x == 0 && b < 1024) g_mw_dbg[b][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define MSTAMP_W(i) do { if (lane == 0 && b < 1024) g_mw_dbg[b][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MSTAMP(i) do { } while (0)
#define MSTAMP_W(i) do { } while (0)
#endif

// phase 3 of one wave: claim the next slot of the ready queue, wait for its fill, estimate, release the fills that waited for it
template <int NS>
__device__ __attribute__((always_inline)) inline void mw_fill_loop(const TeleaWin &win, const TeleaMarchConsts &mc, const uint16_t *fi, const uint16_t *flist,
                                                                   uint32_t *dep, uint16_t *rq, int *ctl, int nfill, int lane)
{
    const int D = mc.range + 1, side = 2 * D + 1, nn = side * side;         // nn <= 121 for range <= 4: two cells per lane
    int noff[2];
    bool non[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) {
        const int j = c2 * 64 + lane;
        non[c2] = j < nn;
        const int dk = j / side - D, dl = j % side - D;
        noff[c2] = non[c2] ? dk * win.ww + dl : 0;
    }
    for (;;) {
        int s = 0;
        if (lane == 0) s = atomicAdd(&ctl[MWC_HEAD], 1);
        s = __builtin_amdgcn_readfirstlane(s);
        if (s >= nfill) break;                   // every fill is queued exactly once: slots [0, nfill) all get an entry
        unsigned k;
        while ((k = ((volatile uint16_t *)rq)[s]) == 0xFFFFu) __builtin_amdgcn_s_sleep(1);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int pi = flist[k];
        telea_fill_known_T<NS>(win, mc, pi, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            if (!non[c2]) continue;
            const unsigned c = fi[pi + noff[c2]];
            if (c > k && c < FI_NOHOLE) {
                if (atomicSub(&dep[c], 1u) == 1u) {
                    const int slot = atomicAdd(&ctl[MWC_TAIL], 1);
                    ((volatile uint16_t *)rq)[slot] = (uint16_t)c;
                }
            }
        }
    }
}

__global__ __launch_bounds__(MW_T) void k_telea_window_mw(float *__restrict__ img_all, const uint8_t *__restrict__ bad_all, const int32_t *__restrict__ box,
                                                          int32_t *__restrict__ fb, int range, int B, int h, int w)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mw_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int P = h * w;
    const int xmin = box[b], ymin = box[B + b], xmax = box[2 * B + b], ymax = box[3 * B + b];
    if (xmin == 0x7f7f7f7f) { if (tid == 0) fb[b] = 0; return; }          // no hole pixel: nothing to inpaint
    const int M = range + 1;
    // window in padded frame coordinates [i0, i1] x [j0, j1], not clipped (cells beyond the image: BORDER)
    const int i0 = ymin + 1 - M, i1 = ymax + 1 + M;
    const int j0 = xmin + 1 - M, j1 = xmax + 1 + M;
    const int wh = i1 - i0 + 1, ww = j1 - j0 + 1;
    const int cells = wh * ww;
    if (cells > MW_CELLS) { if (tid == 0) fb[b] = 1; return; }
    float *img = img_all + (size_t)b * P;
    const uint8_t *bad = bad_all + (size_t)b * P;

    unsigned long long *qo = (unsigned long long *)mw_lds;       // [MW_QCAP] outside pass; phase 3: dependence counters u32 [MW_FILLS]
    unsigned long long *qa = qo + MW_QCAP;                       // [MW_QCAP] ordering pass; phase 3: ready queue u16 [MW_FILLS]
    float *t = (float *)(qa + MW_QCAP);                          // [MW_CELLS]
    float *im = t + MW_CELLS;                                    // [MW_CELLS]
    uint16_t *fi = (uint16_t *)(im + MW_CELLS);                  // [MW_CELLS]
    uint16_t *flist = fi + MW_CELLS;                             // [MW_FILLS]
    uint32_t *hotL = (uint32_t *)(flist + MW_FILLS);             // [2][64]
    int *ctl = (int *)(hotL + 128);                              // [MWC_N]
    uint8_t *f = (uint8_t *)(ctl + MWC_N);                       // [MW_CELLS]
    const uint32_t mg_ww = (uint32_t)(0x100000000ull / (unsigned)ww) + 1u;       // li / ww == umulhi(li, mg_ww) for li < 2^16
    MSTAMP(0);

    // ---- phase 0: window load (hole / border bits, T = 1e6, image), four cells per thread in flight
    if (tid < MWC_N) ctl[tid] = 0;
    for (int base = 0; base < cells; base += MW_T * 4) {
        float v[4];
        uint8_t bd[4];
        bool interior[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int li = base + k * MW_T + tid;
            const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
            const int gi = i0 + r, gj = j0 + cc;
            interior[k] = li < cells && gi >= 1 && gi <= h && gj >= 1 && gj <= w;
            const size_t gp = interior[k] ? (size_t)(gi - 1) * w + (gj - 1) : 0;
            v[k] = img[gp];
            bd[k] = bad[gp];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int li = base + k * MW_T + tid;
            if (li >= cells) continue;
            const bool hole = interior[k] && bd[k];
            im[li] = interior[k] ? v[k] : 0.f;
            f[li] = !interior[k] ? W_BORDER : hole ? W_HOLE : (uint8_t)0;
            fi[li] = hole ? FI_INSIDE : FI_NOHOLE;
            t[li] = 1.0e6f;
        }
    }
    __syncthreads();
    // ring = within Chebyshev `range` of the hole (separable: rows, then columns); band = 4-neighbours of the hole
    {
        const bool unrolled = range <= MW_RING_U;
        for (int li = tid; li < cells; li += MW_T) {
            const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
            const uint8_t me = f[li];
            uint8_t a = 0;
            if (unrolled) {
#pragma unroll
                for (int d = -MW_RING_U; d <= MW_RING_U; d++) {
                    const int c2 = cc + d;
                    const bool in = d >= -range && d <= range && c2 >= 0 && c2 < ww;
                    const uint8_t v = f[in ? li + d : li];
                    a |= in ? v : (uint8_t)0;
                }
            } else {
                const int lo = max(0, cc - range), hi = min(ww - 1, cc + range);
                for (int c2 = lo; c2 <= hi; c2++) a |= f[r * ww + c2];
            }
            if (a & W_HOLE) f[li] = me | W_ROW;           // neighbours only look at bit 6, which this never changes
        }
        __syncthreads();
        for (int li = tid; li < cells; li += MW_T) {
            const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
            const uint8_t me = f[li];
            // window edge cells are at distance range+1 from the hole: never band, their neighbours are not needed
            const uint8_t nb4 = f[cc > 0 ? li - 1 : li] | f[cc < ww - 1 ? li + 1 : li] | f[r > 0 ? li - ww : li] | f[r < wh - 1 ? li + ww : li];
            uint8_t a = 0;
            if (unrolled) {
#pragma unroll
                for (int d = -MW_RING_U; d <= MW_RING_U; d++) {
                    const int r2 = r + d;
                    const bool in = d >= -range && d <= range && r2 >= 0 && r2 < wh;
                    const uint8_t v = f[in ? li + d * ww : li];
                    a |= in ? v : (uint8_t)0;
                }
            } else {
                const int lo = max(0, r - range), hi = min(wh - 1, r + range);
                for (int r2 = lo; r2 <= hi; r2++) a |= f[r2 * ww + cc];
            }
            // (reads of this sweep look at W_ROW / W_HOLE only, bits 5 and 6, which the byte stores below never change)
            if (me & (W_BORDER | W_HOLE)) continue;                          // hole pixels are KNOWN for the outside pass
            if (nb4 & W_HOLE) { f[li] = me | W_SEED; t[li] = 0.f; }
            else if (a & W_ROW) f[li] = me | W_INSIDE;                       // writes bits 0-1 only
        }
    }
    __syncthreads();
    MSTAMP(1);

    // ---- phase 1: the two FMM passes side by side
    if (wave == 0) {
        WQ q;
        q.e = qo; q.hotL = hotL; q.ovf = 0; q.cap = MW_QCAP;
        wq_init(q);
        FmmFlagState st{t, f};
        unsigned long long np = 0, ns = 0;
        telea_fmm_pass(st, q, f, cells, ww, lane, np, ns);
        if (q.ovf && lane == 0) ctl[MWC_FAIL] = 1;
        MSTAMP_W(2);
#ifdef VISTAF_DEBUG
        if (lane == 0 && b < 1024) g_mw_dbg[b][8] = (np << 32) | ns;
#endif
    } else if (wave == 1) {
        WQ q;
        q.e = qa; q.hotL = hotL + 64; q.ovf = 0; q.cap = MW_QCAP;
        wq_init(q);
        FmmOrderState st{t, fi, flist, 0, MW_FILLS};
        unsigned long long np = 0, ns = 0;
        telea_fmm_pass(st, q, f, cells, ww, lane, np, ns);
        if (lane == 0) {
            ctl[MWC_NFILL] = st.n;
            if (q.ovf || st.n > MW_FILLS) ctl[MWC_FAIL] = 1;
        }
        MSTAMP_W(3);
#ifdef VISTAF_DEBUG
        if (lane == 0 && b < 1024) g_mw_dbg[b][9] = (np << 32) | ns;
#endif
    }
    __syncthreads();
    MSTAMP(4);
    if (ctl[MWC_FAIL]) { if (tid == 0) fb[b] = 1; return; }       // nothing has been written back: the single-wave tier marches this frame
    const int nfill = ctl[MWC_NFILL];

    // ---- phase 2: negate T where the outside pass ran, the march's states (hole = INSIDE, rest KNOWN), dependence counters
    uint32_t *dep = (uint32_t *)qo;
    uint16_t *rq = (uint16_t *)qa;
    for (int li = tid; li < cells; li += MW_T) {
        const uint8_t v = f[li];
        if ((v & W_ST) == W_CHANGE) t[li] = -t[li];
        f[li] = (uint8_t)((v & (W_SEED | W_HOLE | W_BORDER)) | ((v & W_HOLE) ? W_INSIDE : W_KNOWN));
    }
    for (int k = tid; k < nfill; k += MW_T) rq[k] = 0xFFFFu;
    __syncthreads();
    {
        const int D = range + 1;
        for (int k = tid; k < nfill; k += MW_T) {
            const int pi = flist[k];
            uint32_t cnt = 0;
            for (int dk = -D; dk <= D; dk++) {
                const uint16_t *row = fi + pi + dk * ww;
                for (int dl = -D; dl <= D; dl++) cnt += row[dl] < (unsigned)k ? 1u : 0u;       // FI_NOHOLE / FI_INSIDE never are
            }
            dep[k] = cnt;
            if (cnt == 0) { const int slot = atomicAdd(&ctl[MWC_TAIL], 1); rq[slot] = (uint16_t)k; }
        }
    }
    __syncthreads();
    MSTAMP(5);

    // ---- phase 3: the estimates
    {
        TeleaWin win;
        win.t = t; win.im = im; win.f = f; win.ww = ww;
        const TeleaMarchConsts mc = telea_march_consts(lane, ww, range);
        switch (mc.ndisc) {
        case 5: mw_fill_loop<5>(win, mc, fi, flist, dep, rq, ctl, nfill, lane); break;
        case 13: mw_fill_loop<13>(win, mc, fi, flist, dep, rq, ctl, nfill, lane); break;
        case 29: mw_fill_loop<29>(win, mc, fi, flist, dep, rq, ctl, nfill, lane); break;
        default: mw_fill_loop<49>(win, mc, fi, flist, dep, rq, ctl, nfill, lane); break;       // range 4 (the launcher admits ranges 1..4 only)
        }
    }
    __syncthreads();
    MSTAMP(6);

    // ---- phase 4: write back the hole pixels
    for (int li = tid; li < cells; li += MW_T) {
        if (!(f[li] & W_HOLE)) continue;
        const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
        img[(size_t)(i0 + r - 1) * w + (j0 + cc - 1)] = im[li];
    }
    if (tid == 0) fb[b] = 0;
    MSTAMP(7);
#ifdef VISTAF_DEBUG
    if (tid == 0 && b < 1024) g_mw_dbg[b][10] = ((unsigned long long)nfill << 32) | (unsigned)cells;
#endif
}

bool inpaint_window_mw_supported(int range) { return range >= 1 && range <= 4; }

void launch_telea_window_mw(float *img, const uint8_t *bad, const int32_t *box, int32_t *fb, int range, int B, int h, int w, hipStream_t st)
{
    static DynLdsOnce lds_once;
    ensure_dyn_lds(lds_once, (const void *)k_telea_window_mw, 160 * 1024);
    hipLaunchKernelGGL(k_telea_window_mw, dim3(B), dim3(MW_T), MW_LDS, st, img, bad, box, fb, range, B, h, w);
}

#ifdef VISTAF_DEBUG
void telea_window_mw_debug_dump(int B)
{
    static unsigned long long hbuf[1024][16];
    if (hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_mw_dbg), sizeof(hbuf)) != hipSuccess) return;
    int worst = 0;
    double mean = 0;
    for (int b = 0; b < B && b < 1024; b++) {
        if (hbuf[b][7] - hbuf[b][0] > hbuf[worst][7] - hbuf[worst][0]) worst = b;
        mean += (double)(hbuf[b][7] - hbuf[b][0]) / B;
    }
    for (int b : {0, worst}) {
        unsigned long long *x = hbuf[b];
        printf("[telea mw dbg] frame %d cycles: load+ring %llu | outside pass %llu (%llu pops in %llu steps) | ordering pass %llu (%llu pops in %llu steps) | "
               "phase 1 %llu | negate+counters %llu | fills %llu | write back %llu | fills %llu cells %llu | mean total over frames %.0f\n",
               b, x[1] - x[0], x[2] - x[1], x[8] >> 32, x[8] & 0xffffffffull, x[3] - x[1], x[9] >> 32, x[9] & 0xffffffffull, x[4] - x[1], x[5] - x[4],
               x[6] - x[5], x[7] - x[6], x[10] >> 32, x[10] & 0xffffffffull, mean);
    }
}
#endif

}  // namespace vf
