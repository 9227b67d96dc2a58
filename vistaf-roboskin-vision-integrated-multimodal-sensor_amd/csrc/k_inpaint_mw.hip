// cv2.inpaint(INPAINT_TELEA) on the frame's hole window -- one WORKGROUP of 16 waves per frame (shape_ftp.py:652-666).
//
// The single-wave window kernel (k_inpaint_win.hip) replays cv::inpaint pop by pop.  Two facts let a whole workgroup work on one frame
// without changing a bit of the result:
//
//  (1) What the march does in which order depends on the mask only: FastMarching_solve reads the flags and T, never the image.  So T of
//      every hole pixel and the sequence of the fills can be fixed first (an FMM pass without estimator, like the outside pass), and the
//      estimates then run as a DATAFLOW: a fill reads flags, T and image values within Chebyshev distance range + 1 of its pixel, so two
//      fills in reach of each other must run in march order, and fills farther apart commute.
//
//  (2) An FMM pass itself is a sequence of GENERATIONS.  A pop at T_p only pushes values T >= T_p + 0.7071 (every non-INSIDE 4-neighbour
//      of a pixel that is filled now is still in the queue -- had it popped earlier, that pop would have filled the pixel -- so both
//      arguments of the solve are >= T_p, and min + (|d| + sqrt(2 - d^2)) / 2 >= min + sqrt(1/2)).  Hence the entries with
//      T < T_head + 0.70 pop next, in their present (T, push order), whatever is pushed meanwhile: a generation.  Inside a generation
//      the order is known up front, a pop writes within Manhattan distance 1 of its pixel and reads within 2, so pops at distance >= 4
//      commute and the others run in queue order -- again a dataflow.  The pushes of a generation carry (generation, parent's rank,
//      neighbour) as push order, which is the order the one-at-a-time loop would have pushed them in.
//      The bench frames: 6 generations for the outside pass, 7-9 for the hole, ~2 000 pops each in ~160 / ~230 dependence levels
//      (the band pixels' raster-order chains make up 100 of them), ~1 400 fills in ~200 levels (tools/telea_dag.py).
//
// Phases (all 16 waves unless noted): window flags + ring | outside pass (icvCalcFMM with negate), generation by generation |
// the march's ordering pass over the hole, generation by generation | image load, negate, dependence counters of the fills |
// the estimates (telea_fill_known_T: the single-wave fill block minus solve and push) from a ready queue | write back.
// Frames whose window, queues or lists do not fit are flagged in fb[] for the full-size single-wave tier (k_telea_window_retry).
#include <cstdio>
#include "kernels.hpp"
#include <type_traits>
#include "telea_common.hpp"

namespace vf {

constexpr int MW_WAVES = 16;
constexpr int MW_T = MW_WAVES * 64;
constexpr int MW_CELLS = 10752;     // window cells: 11 B each (T f32, image f32, fill number u16, flags u8)
constexpr int MW_FILLS = 4096;      // hole pixels per frame
constexpr int MW_RING_U = 6;
constexpr int GP_POOL = 4096;       // pushes of one FMM pass (the pool is append-only: popped entries are blanked)
constexpr int GP_GEN = 2048;        // entries of one generation
constexpr int GP_MAXGEN = 32;
constexpr int GP_CNT = 256;         // chunk counts / bitmap prefix
// fill-number plane: hole pixel not filled yet / not a hole pixel / filled (number pending) / queued in the current generation with rank r
constexpr uint16_t FI_INSIDE = 0x3FFFu, FI_NOHOLE = 0x3FFEu, FI_FILLED = 0x3FFDu, FI_PEND = 0x8000u;
enum { MWC_FAIL = 0, MWC_NFILL, MWC_HEAD, MWC_TAIL, MWC_POOLN, MWC_GENN, MWC_TMIN, MWC_N = 16 };
constexpr size_t MW_REGION_A = (size_t)GP_POOL * 8;          // FMM: pool; fills: dependence counters u32 + ready queue u32
constexpr size_t MW_LDS = MW_REGION_A + (size_t)MW_CELLS * 11 + (size_t)MW_FILLS * 2 + GP_CNT * 4 + MWC_N * 4;
static_assert(MW_LDS <= 160 * 1024, "one CU's LDS");
static_assert((size_t)MW_FILLS * 8 <= MW_REGION_A, "dependence counters and ready queue of the fills");
static_assert((size_t)GP_GEN * (8 + 4) + GP_GEN * 4 / 8 <= (size_t)MW_CELLS * 4, "generation scratch lives in the image plane until the image is loaded");
static_assert(MW_FILLS < FI_FILLED && GP_GEN <= 0x800 && MW_CELLS <= 0x4000 && MW_CELLS / 64 <= GP_CNT && GP_GEN * 4 / 32 <= GP_CNT, "field widths");

#ifdef VISTAF_DEBUG
__device__ unsigned long long g_mw_dbg[1024][16];
#define MSTAMP(i) do { if (threadIdx.x == 0 && b < 1024) g_mw_dbg[b][i] = __builtin_amdgcn_s_memtime(); } while (0)
// cycle sums over the generations of both passes: [11] marks + counters, [12] pops, [13] numbering, [14] next generation (min, select, sort), [15] generations
#define GACC(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_mw_dbg[blockIdx.x][i] += t_ - gstamp_; gstamp_ = t_; } } while (0)
#define GSTART() unsigned long long gstamp_ = __builtin_amdgcn_s_memtime()
#else
#define MSTAMP(i) do { } while (0)
#define GACC(i) do { } while (0)
#define GSTART() do { } while (0)
#endif

struct GenScratch {
    unsigned long long *pool;       // [GP_POOL] (T bits | generation | parent's rank | neighbour | cell), blank = ~0
    unsigned long long *gen;        // [GP_GEN] the current generation in pop order
    uint32_t *dep;                  // [GP_GEN] earlier entries of the generation within Manhattan distance 3 that have not popped yet
    uint32_t *bitmap;               // [GP_GEN * 4 / 32] fills of the generation by (rank, neighbour)
    int *cnt;                       // [GP_CNT]
    int *ctl;
};
__device__ inline unsigned long long gp_key(float T, int g, int rank, int nb, int cell)
{
    return ((unsigned long long)__float_as_uint(T) << 32) | ((unsigned long long)(((uint32_t)g << 27) | ((uint32_t)rank << 16) | ((uint32_t)nb << 14) | (uint32_t)cell));
}
__device__ inline int wave_excl_scan(int v, int lane, int &total)
{
    int s = v;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(s, o, 64); if (lane >= o) s += u; }
    total = __shfl(s, 63, 64);
    return s - v;
}

// The pops of one generation on one wave.  A pop needs 16 lanes (4 neighbours x 4 quadrants), so the wave holds FOUR entries at a time, one
// per 16-lane group: 64 entries of the generation are held by the workgroup's waves, claimed in pop order (the first entry that has not
// popped yet is therefore always held and has no open dependence: the loop makes progress whatever the others wait for).  Every turn a group
// without an entry claims the next one, every group looks at its entry's counter of earlier entries in reach that have not popped yet, and the
// groups whose counter is 0 pop together -- two such entries are never in reach of each other (the later one would be waiting for the earlier).
// ORDER: the march's ordering pass (states in the fill-number plane), else the outside pass (states in the flag bytes).
template <bool ORDER>
__device__ __attribute__((always_inline)) inline void gp_pop_loop(const GenScratch &S, float *t, uint8_t *f, uint16_t *fi, int M, int g, int wh, int ww,
                                                                  uint32_t mg_ww, int lane)
{
    const TeleaOutsideConsts oc = telea_outside_consts(lane, ww);       // per 16-lane group: 4 neighbours x 4 quadrants
    const int nbi = (lane >> 2) & 3, grp = lane >> 4, li = lane & 15;
    // the 24 cells within Manhattan distance 3: lane li of a group takes cells li and li + 16
    int ndy[2], ndx[2];
    bool non[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; c2++) {
        int i = li + 16 * c2;
        non[c2] = i < 24;
        ndy[c2] = 0; ndx[c2] = 0;
        const int rows[7] = {1, 3, 5, 6, 5, 3, 1};
        int dy = -3;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            if (i >= 0 && i < rows[k]) {
                const int rad = 3 - (dy < 0 ? -dy : dy);
                int dx = i - rad;
                if (dy == 0 && dx >= 0) dx++;              // skip the centre
                ndy[c2] = dy; ndx[c2] = dx;
            }
            i -= rows[k];
            dy++;
        }
    }
    const unsigned long long below_grp = (1ull << (16 * grp)) - 1ull;
    int *ctl = S.ctl;
    int r = -1, p = 0;                  // this group's entry (rank in the generation, cell); -1: none
    bool spent = false;                 // the generation has no entry left to claim
    // a pop's pushes wait in registers until the wave has nothing to pop: queue insertions and claims (LDS atomics with a return value) are
    // housekeeping, the time between "the counter reads 0" and "the dependants' counters are decremented" is the workgroup's critical path
    bool plead = false;
    float pdist = 0.f;
    int ppn = 0, pr = 0;
    for (;;) {
        const bool ready = r >= 0 && __hip_atomic_load(&S.dep[r >= 0 ? r : 0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u;
        if (__ballot(ready)) {
            if (ready) {
                const int pn = p + oc.dn, p1 = pn + oc.d1, p2 = pn + oc.d2;
                const int py = (int)__umulhi((uint32_t)p, mg_ww), px = p - py * ww;
                const bool nin0 = non[0] && py + ndy[0] >= 0 && py + ndy[0] < wh && px + ndx[0] >= 0 && px + ndx[0] < ww;
                const bool nin1 = non[1] && py + ndy[1] >= 0 && py + ndy[1] < wh && px + ndx[1] >= 0 && px + ndx[1] < ww;
                // every read of the pop in one round: the neighbour's state, the two arguments of its quadrant (an address outside the window's
                // planes returns junk nobody looks at), the rank marks around p (entries that wait for this pop keep their mark until they pop)
                const float a11 = t[p1], a22 = t[p2];
                bool in0, in1, in2;
                if (ORDER) { const uint16_t s0 = fi[pn], s1 = fi[p1], s2 = fi[p2]; in0 = s0 == FI_INSIDE; in1 = s1 == FI_INSIDE; in2 = s2 == FI_INSIDE; }
                else { const uint8_t s0 = f[pn], s1 = f[p1], s2 = f[p2]; in0 = (s0 & W_ST) == W_INSIDE; in1 = (s1 & W_ST) == W_INSIDE; in2 = (s2 & W_ST) == W_INSIDE; }
                const unsigned v0 = fi[nin0 ? p + ndy[0] * ww + ndx[0] : p], v1 = fi[nin1 ? p + ndy[1] * ww + ndx[1] : p];
                float dist = wn_solve(a11, a22, !in1, !in2);
                {
                    float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0xB1, 0xf, 0xf, false)); dist = o < dist ? o : dist;
                    o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dist), 0x4E, 0xf, 0xf, false)); dist = o < dist ? o : dist;
                }
                if (li == 0) {                          // the entry leaves the queue (this also drops its rank mark)
                    if (ORDER) fi[p] = g == 0 ? FI_NOHOLE : FI_FILLED;
                    else { f[p] = (uint8_t)(g == 0 ? (W_SEED | W_CHANGE) : W_CHANGE); fi[p] = FI_NOHOLE; }
                }
                plead = in0 && (li & 3) == 0;
                pdist = dist; ppn = pn; pr = r;
                if (plead) {
                    t[pn] = dist;
                    if (ORDER) fi[pn] = FI_FILLED; else f[pn] = W_BAND;
                }
                // release the later entries in reach.  The stores above and the decrements below are LDS operations of one wave: the LDS
                // unit executes them in issue order, so a wave that sees its counter at 0 sees the stores (no s_waitcnt between them: the
                // compiler only has to keep the order)
                asm volatile("" ::: "memory");
                const unsigned rv0 = v0 & 0x7FFFu, rv1 = v1 & 0x7FFFu;
                if (nin0 && (v0 & FI_PEND) && rv0 > (unsigned)r) __hip_atomic_fetch_sub(&S.dep[rv0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (nin1 && (v1 & FI_PEND) && rv1 > (unsigned)r) __hip_atomic_fetch_sub(&S.dep[rv1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                r = -1;
            }
            continue;                                   // look at the counters again before any housekeeping
        }
        // ---- nothing to pop: housekeeping.  First the pushes of the pops since the last time ...
        const unsigned long long pb = __ballot(plead);
        if (pb) {
            const int cnt = __popcll(pb);
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctl[MWC_POOLN], cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            const bool fits = base + cnt <= GP_POOL;
            if (!fits && lane == 0) ctl[MWC_FAIL] = 1;
            if (plead) {
                if (fits) S.pool[base + __popcll(pb & ((1ull << lane) - 1ull))] = gp_key(pdist, g, pr, nbi, ppn);
                if (ORDER) atomicOr(&S.bitmap[((unsigned)pr * 4 + nbi) >> 5], 1u << (((unsigned)pr * 4 + nbi) & 31));
            }
            plead = false;
        }
        // ... then the next entries for the groups that have none
        const bool want = r < 0 && !spent;
        const unsigned long long wb = __ballot(want && li == 0);
        if (wb) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctl[MWC_HEAD], __popcll(wb));
            base = __builtin_amdgcn_readfirstlane(base);
            if (want) {
                const int mine = base + __popcll(wb & below_grp);
                if (mine < M) { r = mine; p = (int)((uint32_t)S.gen[mine] & 0x3FFFu); }
                else spent = true;
            }
            continue;
        }
        if (!__ballot(r >= 0)) break;                   // every group is out of entries and every push is queued
        __builtin_amdgcn_s_sleep(1);
    }
}

// One FMM pass over the window, generation by generation.  Returns false (uniformly) when a capacity was exceeded.
template <bool ORDER>
__device__ __attribute__((always_inline)) inline bool gp_pass(const GenScratch &S, float *t, uint8_t *f, uint16_t *fi, uint16_t *flist, int cells, int wh, int ww,
                                                              uint32_t mg_ww, int tid, int lane, int wave)
{
    int *ctl = S.ctl;
    // generation 0: the band pixels in raster order (T = 0)
    const int nchunk = (cells + 63) >> 6;
    for (int ch = wave; ch < nchunk; ch += MW_WAVES) {
        const int li = ch * 64 + lane;
        const unsigned long long bal = __ballot(li < cells && (f[li] & W_SEED));
        if (lane == 0) S.cnt[ch] = __popcll(bal);
    }
    if (tid == 0) { ctl[MWC_POOLN] = 0; ctl[MWC_TMIN] = (int)0xFFFFFFFFu; if (ORDER) ctl[MWC_NFILL] = 0; }
    __syncthreads();
    if (wave == 0) {
        int a[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int ch = lane * 4 + k; a[k] = ch < nchunk ? S.cnt[ch] : 0; sum += a[k]; }
        int total;
        int ex = wave_excl_scan(sum, lane, total);
#pragma unroll
        for (int k = 0; k < 4; k++) { const int ch = lane * 4 + k; if (ch < nchunk) S.cnt[ch] = ex; ex += a[k]; }
        if (lane == 0) ctl[MWC_GENN] = total;
    }
    __syncthreads();
    int M = ctl[MWC_GENN];
    if (M > GP_GEN) return false;
    for (int ch = wave; ch < nchunk; ch += MW_WAVES) {
        const int li = ch * 64 + lane;
        const bool sd = li < cells && (f[li] & W_SEED);
        const unsigned long long bal = __ballot(sd);
        if (sd) { const int r = S.cnt[ch] + __popcll(bal & ((1ull << lane) - 1ull)); S.gen[r] = gp_key(0.f, 0, r, 0, li); }
    }
    __syncthreads();

    GSTART();
    for (int g = 0;; g++) {
        // rank marks
        for (int r = tid; r < M; r += MW_T) {
            const int cell = (int)((uint32_t)S.gen[r] & 0x3FFFu);
            fi[cell] = (uint16_t)(FI_PEND | r);
        }
        if (ORDER) for (int i = tid; i < GP_GEN * 4 / 32; i += MW_T) S.bitmap[i] = 0;
        if (tid == 0) { ctl[MWC_HEAD] = 0; ctl[MWC_TAIL] = 0; ctl[MWC_GENN] = 0; }
        const int pool_n0 = ctl[MWC_POOLN];
        __syncthreads();
        // dependence counters: earlier entries of this generation within Manhattan distance 3
        for (int r = tid; r < M; r += MW_T) {
            const int cell = (int)((uint32_t)S.gen[r] & 0x3FFFu);
            const int py = (int)__umulhi((uint32_t)cell, mg_ww), px = cell - py * ww;
            uint32_t c = 0;
#pragma unroll
            for (int dy = -3; dy <= 3; dy++) {
                const int rad = 3 - (dy < 0 ? -dy : dy);
                const bool yin = py + dy >= 0 && py + dy < wh;
#pragma unroll
                for (int dx = -rad; dx <= rad; dx++) {
                    if (dy == 0 && dx == 0) continue;
                    const bool in = yin && px + dx >= 0 && px + dx < ww;
                    const unsigned v = fi[in ? cell + dy * ww + dx : cell];
                    c += (in && (v & FI_PEND) && (v & 0x7FFFu) < (unsigned)r) ? 1u : 0u;
                }
            }
            S.dep[r] = c;
        }
        __syncthreads();
        GACC(11);
        gp_pop_loop<ORDER>(S, t, f, fi, M, g, wh, ww, mg_ww, lane);
        __syncthreads();
        GACC(12);
        if (ctl[MWC_FAIL]) return false;
        const int pool_n = ctl[MWC_POOLN];
        if (ORDER) {
            // number this generation's fills in push order = by (parent's rank, neighbour)
            if (wave == 0) {
                int a[4], sum = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { a[k] = __popc(S.bitmap[lane * 4 + k]); sum += a[k]; }
                int total;
                int ex = wave_excl_scan(sum, lane, total);
#pragma unroll
                for (int k = 0; k < 4; k++) { S.cnt[lane * 4 + k] = ex; ex += a[k]; }
            }
            __syncthreads();
            const int nf0 = ctl[MWC_NFILL];
            for (int i = pool_n0 + tid; i < pool_n; i += MW_T) {
                const uint32_t lo = (uint32_t)S.pool[i];
                const unsigned bi = (lo >> 14) & 0x1FFFu;                  // rank * 4 + neighbour
                const int num = nf0 + S.cnt[bi >> 5] + __popc(S.bitmap[bi >> 5] & ((1u << (bi & 31)) - 1u));
                if (num < MW_FILLS) flist[num] = (uint16_t)(lo & 0x3FFFu);
            }
            __syncthreads();
            if (tid == 0) ctl[MWC_NFILL] = nf0 + (pool_n - pool_n0);
        }
        GACC(13);
        // the next generation: everything below T_head + 0.70
        {
            uint32_t mn = 0xFFFFFFFFu;
            for (int i = tid; i < pool_n; i += MW_T) { const uint32_t tb = (uint32_t)(S.pool[i] >> 32); mn = tb < mn ? tb : mn; }
            for (int o = 32; o; o >>= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)mn, o, 64); mn = u < mn ? u : mn; }
            if (lane == 0 && mn != 0xFFFFFFFFu) atomicMin((unsigned int *)&ctl[MWC_TMIN], mn);
        }
        __syncthreads();
        const uint32_t tmin = (uint32_t)ctl[MWC_TMIN];
        if (tmin == 0xFFFFFFFFu) break;                                   // queue empty: the pass is complete
        if (g + 1 >= GP_MAXGEN) return false;
        const uint32_t thr = __float_as_uint(__uint_as_float(tmin) + 0.70f);
        for (int i = tid; i < pool_n; i += MW_T) {
            const unsigned long long k = S.pool[i];
            if ((uint32_t)(k >> 32) < thr) {
                const int slot = atomicAdd(&ctl[MWC_GENN], 1);
                if (slot < GP_GEN) S.gen[slot] = k;
                S.pool[i] = ~0ull;
            }
        }
        __syncthreads();
        M = ctl[MWC_GENN];
        if (tid == 0) ctl[MWC_TMIN] = (int)0xFFFFFFFFu;
        if (M > GP_GEN) return false;
        // pop order of the generation: rank = number of smaller keys (keys are unique)
        {
            unsigned long long my[2];
            int c[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < 2; k++) my[k] = tid + k * MW_T < M ? S.gen[tid + k * MW_T] : ~0ull;
            if (M <= MW_T) {
                for (int j = 0; j < M; j++) c[0] += S.gen[j] < my[0] ? 1 : 0;
            } else {
                for (int j = 0; j < M; j++) { const unsigned long long kj = S.gen[j]; c[0] += kj < my[0] ? 1 : 0; c[1] += kj < my[1] ? 1 : 0; }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; k++) if (tid + k * MW_T < M) S.gen[c[k]] = my[k];
        }
        __syncthreads();
        GACC(14);
#ifdef VISTAF_DEBUG
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_mw_dbg[blockIdx.x][15]++;
#endif
    }
    return true;
}

// the estimates on one wave: claim the next slot of the ready queue, wait for its fill, estimate, release the fills that waited for it
template <int NS>
__device__ __attribute__((always_inline)) inline void mw_fill_loop(const TeleaWin &win, const TeleaMarchConsts &mc, const uint16_t *fi, const uint16_t *flist,
                                                                   uint32_t *dep, uint32_t *rq, int *ctl, int nfill, int lane)
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
        unsigned e;                              // fill number | cell << 16
        while ((e = __hip_atomic_load(&rq[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0xFFFFFFFFu) __builtin_amdgcn_s_sleep(1);
        e = (unsigned)__builtin_amdgcn_readfirstlane((int)e);
        const unsigned k = e & 0xFFFFu;
        const int pi = (int)(e >> 16);
        // the fill numbers around pi are read before the estimate (they do not change), the counters of the later ones are decremented
        // right behind its stores: LDS operations of one wave execute in issue order, no wait is needed between them
        const unsigned c0 = non[0] ? fi[pi + noff[0]] : 0u, c1 = non[1] ? fi[pi + noff[1]] : 0u;
        telea_fill_known_T<NS>(win, mc, pi, lane);
        asm volatile("" ::: "memory");
        auto release = [&](unsigned c) {
            if (c > k && c < FI_FILLED) {
                if (atomicSub(&dep[c], 1u) == 1u) {
                    const int slot = atomicAdd(&ctl[MWC_TAIL], 1);
                    __hip_atomic_store(&rq[slot], c | ((unsigned)flist[c] << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        };
        release(c0);
        release(c1);
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

    unsigned char *regA = mw_lds;                                // [MW_REGION_A]
    float *t = (float *)(regA + MW_REGION_A);                    // [MW_CELLS]
    float *im = t + MW_CELLS;                                    // [MW_CELLS]; generation scratch until the image is loaded
    uint16_t *fi = (uint16_t *)(im + MW_CELLS);                  // [MW_CELLS]
    uint16_t *flist = fi + MW_CELLS;                             // [MW_FILLS]
    int *cntv = (int *)(flist + MW_FILLS);                       // [GP_CNT]
    int *ctl = cntv + GP_CNT;                                    // [MWC_N]
    uint8_t *f = (uint8_t *)(ctl + MWC_N);                       // [MW_CELLS]
    GenScratch S;
    S.pool = (unsigned long long *)regA;
    S.gen = (unsigned long long *)im;
    S.dep = (uint32_t *)(S.gen + GP_GEN);
    S.bitmap = S.dep + GP_GEN;
    S.cnt = cntv;
    S.ctl = ctl;
    const uint32_t mg_ww = (uint32_t)(0x100000000ull / (unsigned)ww) + 1u;       // li / ww == umulhi(li, mg_ww) for li < 2^16
    MSTAMP(0);
#ifdef VISTAF_DEBUG
    if (tid == 0 && b < 1024) for (int i = 11; i < 16; i++) g_mw_dbg[b][i] = 0;
#endif

    // ---- window flags (hole / border bits), T = 1e6
    if (tid < MWC_N) ctl[tid] = 0;
    for (int base = 0; base < cells; base += MW_T * 4) {
        uint8_t bd[4];
        bool interior[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int li = base + k * MW_T + tid;
            const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
            const int gi = i0 + r, gj = j0 + cc;
            interior[k] = li < cells && gi >= 1 && gi <= h && gj >= 1 && gj <= w;
            const size_t gp = interior[k] ? (size_t)(gi - 1) * w + (gj - 1) : 0;
            bd[k] = bad[gp];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int li = base + k * MW_T + tid;
            if (li >= cells) continue;
            const bool hole = interior[k] && bd[k];
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

    // ---- the two FMM passes.  They touch disjoint cells apart from the band pixels both start from (ring cells have no hole neighbour,
    // hole pixels no ring neighbour; band pixels keep T = 0), so their order does not matter
    bool ok = gp_pass<false>(S, t, f, fi, flist, cells, wh, ww, mg_ww, tid, lane, wave);
    MSTAMP(2);
    __syncthreads();
    if (ok) ok = gp_pass<true>(S, t, f, fi, flist, cells, wh, ww, mg_ww, tid, lane, wave);
    MSTAMP(3);
    const int nfill = ctl[MWC_NFILL];
    if (!ok || nfill > MW_FILLS) { if (tid == 0) fb[b] = 1; return; }     // nothing has been written back: the single-wave tier marches this frame
    __syncthreads();

    // ---- image into the window; negate T where the outside pass ran; the march's states (hole = INSIDE, rest KNOWN); fill numbers
    uint32_t *dep = (uint32_t *)regA;
    uint32_t *rq = dep + MW_FILLS;
    for (int base = 0; base < cells; base += MW_T * 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int li = base + k * MW_T + tid;
            const bool interior = li < cells && !(f[li < cells ? li : 0] & W_BORDER);
            const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
            v[k] = interior ? img[(size_t)(i0 + r - 1) * w + (j0 + cc - 1)] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int li = base + k * MW_T + tid;
            if (li >= cells) continue;
            im[li] = v[k];
            const uint8_t fv = f[li];
            if ((fv & W_ST) == W_CHANGE) t[li] = -t[li];
            f[li] = (uint8_t)((fv & (W_SEED | W_HOLE | W_BORDER)) | ((fv & W_HOLE) ? W_INSIDE : W_KNOWN));
        }
    }
    for (int k = tid; k < nfill; k += MW_T) { fi[flist[k]] = (uint16_t)k; rq[k] = 0xFFFFFFFFu; }
    if (tid == 0) { ctl[MWC_HEAD] = 0; ctl[MWC_TAIL] = 0; }
    __syncthreads();
    // dependence counters: earlier fills within Chebyshev distance range + 1 (the reach of a fill's reads)
    {
        const int D = range + 1;
        for (int k = tid; k < nfill; k += MW_T) {
            const int pi = flist[k];
            uint32_t cnt = 0;
            for (int dk = -D; dk <= D; dk++) {
                const uint16_t *row = fi + pi + dk * ww;
                for (int dl = -D; dl <= D; dl++) cnt += row[dl] < (unsigned)k ? 1u : 0u;       // FI_NOHOLE is not
            }
            dep[k] = cnt;
            if (cnt == 0) { const int slot = atomicAdd(&ctl[MWC_TAIL], 1); rq[slot] = (uint32_t)k | ((uint32_t)pi << 16); }
        }
    }
    __syncthreads();
    MSTAMP(4);

    // ---- the estimates
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
    MSTAMP(5);

    // ---- write back the hole pixels
    for (int li = tid; li < cells; li += MW_T) {
        if (!(f[li] & W_HOLE)) continue;
        const int r = (int)__umulhi((uint32_t)li, mg_ww), cc = li - r * ww;
        img[(size_t)(i0 + r - 1) * w + (j0 + cc - 1)] = im[li];
    }
    if (tid == 0) fb[b] = 0;
    MSTAMP(6);
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
        if (hbuf[b][6] - hbuf[b][0] > hbuf[worst][6] - hbuf[worst][0]) worst = b;
        mean += (double)(hbuf[b][6] - hbuf[b][0]) / B;
    }
    for (int b : {0, worst}) {
        unsigned long long *x = hbuf[b];
        printf("[telea mw dbg] frame %d cycles: flags+ring %llu | outside pass %llu | ordering pass %llu | image+negate+counters %llu | fills %llu | "
               "write back %llu | fills %llu cells %llu | mean total over frames %.0f || both passes, %llu generation changes: marks+counters %llu | pops %llu | "
               "numbering %llu | min+select+sort %llu\n",
               b, x[1] - x[0], x[2] - x[1], x[3] - x[2], x[4] - x[3], x[5] - x[4], x[6] - x[5], x[10] >> 32, x[10] & 0xffffffffull, mean, x[15], x[11], x[12], x[13],
               x[14]);
    }
}
#endif

}  // namespace vf
