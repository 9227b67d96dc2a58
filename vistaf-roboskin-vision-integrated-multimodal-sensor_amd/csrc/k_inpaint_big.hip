// cv2.inpaint(INPAINT_TELEA) -- the clusters no LDS window can take (shape_ftp.py:652-666; native 1182 x 1182 crops: the saturated crests
// are strips of ~60 x 400-600 pixels, 3-8 k hole pixels each, 20 k cells within reach of the march).
//
// One wave per cluster, as for the LDS windows (k_inpaint_win.hip), and the SAME per-pop code (telea_common.hpp): only the planes are
// somewhere else.  T, the flag bytes and a copy of the image live in global memory, padded by range + 1 cells on every side so that -- like in
// an LDS window -- `cell + offset` is always a valid cell and OpenCV's first / last row / column index shifts are read off BORDER flags;
// the priority queue stays in LDS (128 KB: its pushes and pops are the march's most frequent dependent accesses).  A fill then costs one
// round trip to L2 for all of its reads (they are issued together, telea_pop_march) instead of one per dependent step.
// The planes are shared by all clusters of a frame: clusters touch disjoint cells (k_inpaint_cl.hip), a wave only ever reads cells its own
// cluster owns or cells nobody writes, and a wave sees its own global stores in program order.
#include <type_traits>
#include "kernels.hpp"
#include "telea_common.hpp"

namespace vf {

constexpr int BG_QCAP = 16384;       // queue entries in LDS (128 KB: one march per CU; a batch of native crops has fewer big clusters than the chip has CUs)
constexpr int BG_MAXPAD = 6;         // planes are padded by range + 1 <= 6 cells
constexpr int BG_MAXSLOTS = 32;      // waves per frame

static int bg_slots(int h, int w) { return (int)std::min<size_t>(BG_MAXSLOTS, std::max<size_t>(4, (size_t)(h + 2) * (w + 2) / 8192)); }
// entries of a wave's queue slice in global memory (a power of two), for a cluster whose band can outgrow the LDS queue
static int bg_gq_cap(int h, int w)
{
    const size_t per = (size_t)(h + 2 * BG_MAXPAD) * (w + 2 * BG_MAXPAD) / (size_t)bg_slots(h, w);
    int cap = 1024;
    while ((size_t)cap * 2 <= per) cap *= 2;
    return cap;
}
// T f32 | image f32 | flags u8 over the padded frame, then the queue slices
size_t inpaint_big_scratch_bytes_per_frame(int h, int w)
{
    const size_t en = (size_t)(h + 2 * BG_MAXPAD) * (w + 2 * BG_MAXPAD);
    return en * 9 + (size_t)bg_slots(h, w) * bg_gq_cap(h, w) * 8 + 1024;
}
bool inpaint_big_supported(int range) { return range >= 1 && range + 1 <= BG_MAXPAD; }

// One thread per padded cell: the flag byte an LDS window would hold after its load + ring phases (wn_march), T and the image copy.
// dil = the hole mask of ALL clusters dilated by range + 1 (k_inpaint_cl.hip): where it is clear no hole pixel is within reach.
__global__ __launch_bounds__(256) void k_bg_prep(const float *__restrict__ img_all, const uint8_t *__restrict__ bad_all, const uint8_t *__restrict__ dil_all,
                                                 float *__restrict__ gT, float *__restrict__ gim, uint8_t *__restrict__ gf, int range, int h, int w)
{
    const int M = range + 1, ew = w + 2 * M, eh = h + 2 * M, en = eh * ew, P = h * w;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= en) return;
    const int Y = i / ew, X = i - Y * ew, y = Y - M, x = X - M;
    const uint8_t *bad = bad_all + b * (size_t)P;
    auto hole = [&](int yy, int xx) -> bool { return yy >= 0 && yy < h && xx >= 0 && xx < w && bad[(size_t)yy * w + xx] != 0; };
    uint8_t fv = 0;
    float tv = 1.0e6f, iv = 0.f;
    if (y < 0 || y >= h || x < 0 || x >= w) fv = W_BORDER;
    else {
        const size_t p = (size_t)y * w + x;
        iv = img_all[b * (size_t)P + p];
        if (bad[p]) fv = W_HOLE;                                  // KNOWN for the outside pass
        else if (dil_all[b * (size_t)P + p]) {
            if (hole(y, x - 1) || hole(y, x + 1) || hole(y - 1, x) || hole(y + 1, x)) { fv = W_SEED; tv = 0.f; }     // initial band
            else {
                bool near = false;
                for (int a = -range; a <= range && !near; a++)
                    for (int c = -range; c <= range; c++)
                        if (hole(y + a, x + c)) { near = true; break; }
                if (near) fv = W_INSIDE;                          // ring of the outside pass
            }
        }
    }
    gf[b * (size_t)en + i] = fv;
    gT[b * (size_t)en + i] = tv;
    gim[b * (size_t)en + i] = iv;
}

// the cells of one cluster inside its window (padded coordinates, inclusive)
struct BgWin {
    int i0, i1, j0, j1, nch, total;      // nch = 64-column chunks per row, total = chunks in the window
    int M, ew, h, w, rootp;
    const int32_t *lab;
};
constexpr int BG_SCAN_U = 4;             // chunks whose label / flag loads are in flight together

// fn(cell, flag) on every cell of the cluster, one cell per lane (any order)
template <class F>
__device__ __attribute__((always_inline)) inline void bg_each(const BgWin &W, const uint8_t *f, int lane, F fn)
{
    for (int k0 = 0; k0 < W.total; k0 += BG_SCAN_U) {
        int cell[BG_SCAN_U], lv[BG_SCAN_U];
        uint8_t fl[BG_SCAN_U];
#pragma unroll
        for (int u = 0; u < BG_SCAN_U; u++) {
            const int k = k0 + u, r = k / W.nch, row = W.i0 + r, col = W.j0 + (k - r * W.nch) * 64 + lane;
            const int y = row - W.M, x = col - W.M;
            const bool v = k < W.total && col <= W.j1 && y >= 0 && y < W.h && x >= 0 && x < W.w;
            cell[u] = row * W.ew + col;
            lv[u] = v ? W.lab[(size_t)y * W.w + x] : -1;
            fl[u] = v ? f[cell[u]] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < BG_SCAN_U; u++)
            if (lv[u] == W.rootp) fn(cell[u], fl[u]);
    }
}
// fn(base, pend) for every 64-column chunk of the window that holds seeds (initial band pixels) of the cluster, in raster order:
// pend = the lanes whose cell base + lane is a seed.  The SEED bit of a cell never changes, so the flags of the next chunks may be loaded
// ahead of the pops of this one.
template <class F>
__device__ __attribute__((always_inline)) inline void bg_seed_chunks(const BgWin &W, const uint8_t *f, int lane, F fn)
{
    for (int k0 = 0; k0 < W.total; k0 += BG_SCAN_U) {
        unsigned long long pend[BG_SCAN_U];
        int base[BG_SCAN_U];
        int lv[BG_SCAN_U];
        uint8_t fl[BG_SCAN_U];
#pragma unroll
        for (int u = 0; u < BG_SCAN_U; u++) {
            const int k = k0 + u, r = k / W.nch, row = W.i0 + r, cb = W.j0 + (k - r * W.nch) * 64, col = cb + lane;
            const int y = row - W.M, x = col - W.M;
            const bool v = k < W.total && col <= W.j1 && y >= 0 && y < W.h && x >= 0 && x < W.w;
            base[u] = row * W.ew + cb;
            lv[u] = v ? W.lab[(size_t)y * W.w + x] : -1;
            fl[u] = v ? f[base[u] + lane] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < BG_SCAN_U; u++) pend[u] = __ballot(lv[u] == W.rootp && (fl[u] & W_SEED));
        static_assert(BG_SCAN_U == 4, "the selects below");
#pragma nounroll
        for (int u = 0; u < BG_SCAN_U; u++) {                       // not unrolled: fn is the whole per-pop body of a pass
            const unsigned long long m = u == 0 ? pend[0] : u == 1 ? pend[1] : u == 2 ? pend[2] : pend[3];
            const int bs = u == 0 ? base[0] : u == 1 ? base[1] : u == 2 ? base[2] : base[3];
            if (m) fn(bs, m);
        }
    }
}

// LQ: the queue's sorted run in LDS (the normal case) or in this wave's slice of global memory.  Returns false when the queue overflowed.
template <bool LQ>
__device__ __attribute__((always_inline)) inline bool bg_march(const BgWin &W, float *t, float *im, uint8_t *f, unsigned char *lds, unsigned long long *gq,
                                                              int gq_cap, int lds_cap, int range, int lane)
{
    TeleaWin win;
    win.t = t; win.im = im; win.f = f; win.ww = W.ew;
    WQ q;
    q.e = LQ ? (unsigned long long *)lds : gq;
    q.hotL = (uint32_t *)(lds + (size_t)lds_cap * 8);
    q.cap = LQ ? lds_cap : gq_cap;
    q.ovf = 0;
    wq_init(q);
    // ---- pass 1: outside T field (icvCalcFMM, negate): seeds first in raster order, then the queue, up to four pops per step
    // (telea_pop_outside4: the row / column split of its distance test is exact as long as cell * pitch < 2^32; else one pop per step)
    {
        const TeleaOutsideConsts oc = telea_outside_consts(lane, W.ew);
        if ((unsigned long long)W.ew * W.ew * (unsigned long long)(W.h + 2 * W.M) < 0x100000000ull) {
            FmmFlagState fst{t, f};
            const uint32_t magic = telea_magic_ww(W.ew);
            unsigned long long np = 0, ns = 0;
            bg_seed_chunks(W, f, lane, [&](int base, unsigned long long pend) { telea_fmm_seed_chunk(fst, q, oc, base, pend, W.ew, magic, lane, np, ns); });
            telea_fmm_queue(fst, q, oc, W.ew, magic, lane, np, ns);
        } else {
            auto push1 = [&](float T_, int idx_) { wq_push<false>(q, T_, idx_, lane); };
            bg_seed_chunks(W, f, lane, [&](int base, unsigned long long pend) {
                while (pend && !q.ovf) { const int l = __ffsll((long long)pend) - 1; pend &= pend - 1ull; telea_pop_outside(win, oc, base + l, true, lane, push1); }
            });
            while (!q.ovf) {
                if (q.nh > 0) wq_merge<false>(q, lane);
                if (q.ovf || q.head == q.tail) break;
                const int p = (int)(uint32_t)q.e[q.head];
                q.head++;
                telea_pop_outside(win, oc, p, false, lane, push1);
            }
        }
    }
    if (q.ovf) return false;
    // negate T where the outside pass ran; switch the state bits to the march's flags (hole = INSIDE, rest KNOWN)
    bg_each(W, f, lane, [&](int cell, uint8_t v) {
        if ((v & W_ST) == W_CHANGE) t[cell] = -t[cell];
        f[cell] = (uint8_t)((v & (W_SEED | W_HOLE | W_BORDER)) | ((v & W_HOLE) ? W_INSIDE : W_KNOWN));
    });
    // ---- pass 2: Telea march (icvTeleaInpaintFMM)
    wq_init(q);
    auto push = [&](float T_, int idx_) { wq_push(q, T_, idx_, lane); };
    const TeleaMarchConsts mc = telea_march_consts(lane, W.ew, range);
    auto march = [&](auto small) {
        bg_seed_chunks(W, f, lane, [&](int base, unsigned long long pend) {
            while (pend && !q.ovf) {
                const int l = __ffsll((long long)pend) - 1;
                pend &= pend - 1ull;
                telea_pop_march<decltype(small)::value>(win, mc, base + l, false, lane, push);
            }
        });
        while (!q.ovf) {
            const int p = wq_pop(q);
            if (p < 0) break;
            telea_pop_march<decltype(small)::value>(win, mc, p, true, lane, push);
        }
    };
    switch (mc.ndisc) {
    case 5: march(std::integral_constant<int, 5>{}); break;
    case 13: march(std::integral_constant<int, 13>{}); break;
    case 29: march(std::integral_constant<int, 29>{}); break;
    case 49: march(std::integral_constant<int, 49>{}); break;
    default: march(std::integral_constant<int, 0>{}); break;
    }
    return !q.ovf;
}

// blockIdx.x walks the list of the frame's left-over clusters (k_inpaint_cl.hip), blockIdx.y is the frame
__global__ __launch_bounds__(64) void k_telea_big_clusters(float *__restrict__ img_all, float *gT, float *gim, uint8_t *gf, unsigned long long *gq_all,
                                                           const int32_t *__restrict__ labels_all, const int32_t *__restrict__ list_all,
                                                           const int32_t *__restrict__ count, const int32_t *__restrict__ xmin,
                                                           const int32_t *__restrict__ ymin, const int32_t *__restrict__ xmax,
                                                           const int32_t *__restrict__ ymax, int32_t *status, int range, int h, int w, int gq_cap, int lds_cap, int lds_use)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char bg_lds[];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.y;
    const int M = range + 1, ew = w + 2 * M, eh = h + 2 * M, P = h * w;
    const size_t en = (size_t)eh * ew;
    const int ncl = count[b];
    float *t = gT + b * en, *im = gim + b * en;
    uint8_t *f = gf + b * en;
    float *img = img_all + b * (size_t)P;
    unsigned long long *gq = gq_all + (b * gridDim.x + blockIdx.x) * (size_t)gq_cap;
    for (int c = blockIdx.x; c < ncl; c += gridDim.x) {
        const int rootp = list_all[b * (size_t)P + c];
        const size_t root = b * (size_t)P + rootp;
        BgWin W;
        // pixel (y, x) is cell (y + M, x + M); the window is the bounding box of the cluster's hole pixels grown by M: always inside the padded plane
        W.i0 = ymin[root]; W.i1 = ymax[root] + 2 * M; W.j0 = xmin[root]; W.j1 = xmax[root] + 2 * M;
        W.nch = (W.j1 - W.j0 + 64) / 64; W.total = (W.i1 - W.i0 + 1) * W.nch;
        W.M = M; W.ew = ew; W.h = h; W.w = w; W.rootp = rootp; W.lab = labels_all + b * (size_t)P;
        // Queue entries are band cells: ring cells in pass 1, hole pixels in pass 2.  When even all of them fit the LDS queue it cannot
        // overflow; a cluster with more cells than that keeps its sorted run in this wave's slice of global memory instead, whose overflow is
        // detected when it happens (the band is a front, a small fraction of the cells: status 2 if it ever outgrows the slice).
        int nring = 0, nhole = 0;
        bg_each(W, f, lane, [&](int, uint8_t v) { nring += (v & W_ST) == W_INSIDE; nhole += (v & W_HOLE) != 0; });
        for (int o = 32; o; o >>= 1) { nring += __shfl_xor(nring, o, 64); nhole += __shfl_xor(nhole, o, 64); }
        const int need = max(nring, nhole) + 128;                    // + the hot run and the slack of a merge
        bool ok;
        if (need <= lds_use) ok = bg_march<true>(W, t, im, f, bg_lds, gq, gq_cap, lds_cap, range, lane);
        else ok = bg_march<false>(W, t, im, f, bg_lds, gq, gq_cap, lds_cap, range, lane);
        if (!ok) { if (lane == 0) status[b] = 2; continue; }
        bg_each(W, f, lane, [&](int cell, uint8_t v) {
            if (v & W_HOLE) { const int Y = cell / ew, X = cell - Y * ew; img[(size_t)(Y - M) * w + (X - M)] = im[cell]; }
        });
        __builtin_amdgcn_wave_barrier();
    }
}

// The hole pixels of `bad_big` (clusters too large for an LDS window), one wave per cluster, up to 32 waves per frame.
void launch_inpaint_big_clusters(float *img, const uint8_t *bad_big, int range, void *scratch, int32_t *status, const ClusterPlanes &left, int B, int h,
                                 int w, hipStream_t st, bool lds_queue)
{
    const int M = range + 1;
    const size_t en = (size_t)(h + 2 * M) * (w + 2 * M);
    float *gT = (float *)scratch;
    float *gim = gT + (size_t)B * en;
    uint8_t *gf = (uint8_t *)(gim + (size_t)B * en);
    unsigned long long *gq = (unsigned long long *)((((uintptr_t)(gf + (size_t)B * en)) + 255) & ~(uintptr_t)255);
    const int nslot = bg_slots(h, w), gq_cap = bg_gq_cap(h, w);
    hipLaunchKernelGGL(k_bg_prep, dim3((unsigned)((en + 255) / 256), B), dim3(256), 0, st, img, bad_big, left.dil, gT, gim, gf, range, h, w);
    // LDS queue: 16384 entries (128 KB, one march per CU) while the batch has fewer big clusters than the chip has CUs -- a native crop has
    // about ten --, 8192 (two marches per CU) for larger batches; a cluster whose cell counts exceed the queue takes the global slice
    const int lds_cap = B <= 16 ? BG_QCAP : BG_QCAP / 2;
    static DynLdsOnce lds_once;
    ensure_dyn_lds(lds_once, (const void *)k_telea_big_clusters, BG_QCAP * 8 + 256);
    hipLaunchKernelGGL(k_telea_big_clusters, dim3(nslot, B), dim3(64), (size_t)lds_cap * 8 + 256, st, img, gT, gim, gf, gq, left.labels, left.list, left.count,
                       left.xmin, left.ymin, left.xmax, left.ymax, status, range, h, w, gq_cap, lds_cap, lds_queue ? lds_cap : 0);
}

}  // namespace vf
