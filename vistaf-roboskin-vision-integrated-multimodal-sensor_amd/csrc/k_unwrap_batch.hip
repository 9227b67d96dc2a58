// k_unwrap_flood_batch: the growth loop of unwrap_quality_guided (shape_ftp.py:1043-1080), several pops per
// step.  Same contract as k_unwrap_flood_hot (k_unwrap_hot.hip): one wavefront per frame, padded uint16 rank
// plane in LDS (0 outside the mask, 1 visited, 2 frontier, >= 3 rank code), parents written in padded index
// space, frontier = sorted HOT register list (<= 64 entries, code << 16 | pixel) + COLD bitmap over rank codes.
//
// The reference pops ONE pixel at a time, but measured on this path 95 % of the pixels a pop adds to the
// frontier rank below the 64 best frontier entries, and those best entries are scattered along the frontier.
// So the next K = 8 pops are usually known in advance and do not interact.  A step examines the 8 best HOT
// entries at once (lane = candidate * 8 + neighbour) and commits the longest prefix that provably pops in
// that order with the same neighbour states as the one-at-a-time loop:
//   (i)  a candidate within Chebyshev distance 2 of an earlier candidate ends the prefix before it (it
//        could see the earlier one as a visited neighbour, or share a fresh neighbour with it);
//   (ii) a fresh neighbour that outranks the last candidate may have to pop before a later candidate: the
//        prefix ends with the candidate that produced it and the entry goes into HOT.
// Everything else about a pop (parent = lexicographically smallest visited neighbour, fresh neighbours to the
// frontier) touches only the candidate's own 3x3 block, so the committed pops are independent.
#include <cstdio>
#include "kernels.hpp"

namespace vf {

constexpr int BT_NW = 1024;    // 64-bit words of the cold bitmap (codes < 65536)
constexpr int BT_K = 8;        // candidates per step
#ifdef VISTAF_DEBUG
__device__ unsigned long long g_batch_dbg[8];      // frame 0: steps, candidates, commits, steps cut by (i) / (ii), refills, HOT inserts
#define BT_COUNT(i, v) do { if (b == 0 && lane == 0) g_batch_dbg[i] += (unsigned long long)(v); } while (0)
#else
#define BT_COUNT(i, v) do { } while (0)
#endif

__device__ inline uint32_t bt_shr1(uint32_t v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false); }
__device__ inline uint32_t bt_bperm(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts)
__device__ inline int bt_scan_add(int v)
{
#define BT_STEP(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, true);
    BT_STEP(0x111, 0xf) BT_STEP(0x112, 0xf) BT_STEP(0x114, 0xf) BT_STEP(0x118, 0xf)
#undef BT_STEP
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast31 into rows 2 and 3
    return v;
}

__global__ __launch_bounds__(64) void k_unwrap_flood_batch(const uint16_t *__restrict__ rank_all, const int32_t *__restrict__ seed_in,
                                                           const uint32_t *__restrict__ inv_all, size_t inv_stride, int32_t *__restrict__ ppar_all,
                                                           size_t gstride, uint32_t *__restrict__ order_all, size_t ostride, int h, int w, uint32_t magic, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    const int W2 = w + 2, EN = (h + 2) * W2;
    const int EN8 = (EN + 7) & ~7;
    uint32_t *order = order_all + b * ostride;                           // pop record: parent << 16 | pixel (padded indices); order[ostride-1] = count
    int npop = 0;
    uint16_t *kp = (uint16_t *)lds_raw;                                  // [EN8] pixel state / rank code
    unsigned long long *L0 = (unsigned long long *)(kp + EN8);           // [BT_NW] cold bitmap over codes
    unsigned long long *L1 = L0 + BT_NW;                                 // [16]    one bit per L0 word
    uint32_t *stage = (uint32_t *)(L1 + 16);                             // [64]    refill staging
    uint16_t *icache = (uint16_t *)(stage + 64);                         // [256]   slice of inv just below the last refill (see refill)
    int32_t *ppar = ppar_all + b * gstride;
    const uint16_t *rk = rank_all + b * (size_t)EN8;
    const uint32_t *inv = inv_all + b * inv_stride;                         // inv[rank] = padded pixel index

    {
        const uint4 *src = (const uint4 *)rk;
        uint4 *dst = (uint4 *)kp;
        int nv = EN >> 3;
        for (int i = lane; i < nv; i += 64) dst[i] = src[i];
        for (int p = (nv << 3) + lane; p < EN; p += 64) kp[p] = rk[p];
    }
    for (int i = lane; i < BT_NW + 16; i += 64) L0[i] = 0ull;
    for (int p = lane; p < EN; p += 64) ppar[p] = -1;
    __syncthreads();
    const int seed = __builtin_amdgcn_readfirstlane(seed_in[b]);
    if (seed < 0) { if (lane == 0) order[ostride - 1] = 0u; return; }   // empty mask (shape_ftp.py:1047-1048)
    const int ci = lane >> 3, n = lane & 7;                              // candidate slot, neighbour slot
    int doff;
    {
        int l = n < 4 ? n : n + 1;
        doff = (l / 3 - 1) * W2 + (l % 3 - 1);
    }
    // the seed is the first frontier entry; its own parent is itself (shape_ftp.py:1051)
    uint32_t hot = 0;            // sorted descending over lanes 0..H-1, 0 elsewhere
    {
        uint32_t sc = kp[seed];
        if (lane == 0) { hot = (sc << 16) | (uint32_t)seed; ppar[seed] = seed; }
    }
    int H = 1;                   // entries in HOT (wave-uniform)
    uint32_t tailv = (uint32_t)__builtin_amdgcn_readfirstlane((int)hot);   // hot entry of lane H-1, 0 when H == 0
    bool cold_any = false;       // the cold bitmap may be non-empty
    // inv prefetch: the top of the frontier is dense in code space, so the next refill mostly pulls codes just below the lowest
    // code of this one.  Their pixel indices are fetched now (4 per lane, consumed at the next refill through an LDS slice), which
    // takes the global-memory round trip out of the refill.
    uint32_t pf[4] = {0u, 0u, 0u, 0u};
    int pf_lo = 0, pf_n = 0;          // codes [pf_lo, pf_lo + pf_n) are in flight / cached (wave-uniform)

    for (;;) {
        // ---- refill: HOT holds fewer than K entries: append the top cold codes (already in descending order)
        if (H < BT_K && cold_any) {
            BT_COUNT(5, 1);
            unsigned long long l1 = lane < 16 ? L1[lane] : 0ull;
            unsigned long long nz = __ballot(l1 != 0ull);
            if (nz == 0ull) cold_any = false;
            else {
                int top1 = 63 - __clzll((long long)nz);
                uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)l1, top1);
                uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(l1 >> 32), top1);
                unsigned long long w1 = ((unsigned long long)hi << 32) | lo;
                const int wtop = top1 * 64 + (63 - __clzll((long long)w1));
                const int wi = wtop - lane;
                const unsigned long long word = wi >= 0 ? L0[wi] : 0ull;
                const int cnt = (int)__popcll(word);
                const int incl = bt_scan_add(cnt);
                const int pre = incl - cnt;
                const int room = 64 - H;
                int take = room - pre;
                take = take < 0 ? 0 : (take > cnt ? cnt : take);
                int total = __builtin_amdgcn_readlane(incl, 63);
                total = total > room ? room : total;
                // one bitmap word at a time, one lane per BIT: the j-th highest set bit goes to stage[pre + j]
                unsigned long long wm = __ballot(take > 0);
                const int bpos = 63 - lane;
                while (wm) {
                    const int wl = __ffsll((long long)wm) - 1;
                    wm &= wm - 1ull;
                    const uint32_t xlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)word, wl);
                    const uint32_t xhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(word >> 32), wl);
                    const unsigned long long x = ((unsigned long long)xhi << 32) | xlo;
                    const int xpre = __builtin_amdgcn_readlane(pre, wl);
                    const int xtake = __builtin_amdgcn_readlane(take, wl);
                    const int xwi = wtop - wl;
                    const bool set = (x >> bpos) & 1ull;
                    const int above = (int)__popcll((x >> bpos) >> 1);
                    const bool tk = set && above < xtake;
                    if (tk) stage[xpre + above] = (uint32_t)(xwi * 64 + bpos);
                    const unsigned long long taken = __builtin_bitreverse64(__ballot(tk));     // lane L <-> bit 63 - L
                    const unsigned long long rem = x & ~taken;
                    if (lane == 0) {
                        L0[xwi] = rem;
                        if (rem == 0ull) atomicAnd(&L1[xwi >> 6], ~(1ull << (xwi & 63)));
                    }
                }
                const int k = lane - H;
                const bool mine = k >= 0 && k < total;
                const uint32_t code = mine ? stage[k] : 3u;
                // the slice prefetched at the previous refill lands in LDS now
#pragma unroll
                for (int j = 0; j < 4; j++) icache[lane * 4 + j] = (uint16_t)pf[j];
                __builtin_amdgcn_wave_barrier();
                const bool hit = mine && (int)code >= pf_lo && (int)code < pf_lo + pf_n;
                uint32_t idx = hit ? (uint32_t)icache[(int)code - pf_lo] : 0u;
                if (mine && !hit) idx = inv[code - 3u];
                hot = mine ? ((code << 16) | idx) : hot;
                H += total;
                tailv = (uint32_t)__builtin_amdgcn_readlane((int)hot, H - 1);
                // prefetch the 256 codes below the lowest one pulled
                const int clow = (int)(tailv >> 16);
                pf_lo = clow - 256 < 3 ? 3 : clow - 256;
                pf_n = clow - pf_lo;
#pragma unroll
                for (int j = 0; j < 4; j++) { const int cc = pf_lo + lane * 4 + j; pf[j] = cc < clow ? inv[cc - 3] : 0u; }
            }
        }
        if (H == 0) break;                                               // frontier exhausted

        // ---- examine the K best entries: lane = candidate * 8 + neighbour
        const int mav = H < BT_K ? H : BT_K;
        const uint32_t c = bt_bperm(hot, ci);                             // this lane's candidate
        const uint32_t co = bt_bperm(hot, n);                             // candidate n (for the pairwise distance test)
        const bool valid = ci < mav;
        const int idx = (int)(c & 0xffffu);
        const int np = idx + doff;
        const uint32_t v = valid ? kp[np] : 0u;
        const int oi = (int)(co & 0xffffu);
        const int y = (int)__umulhi((uint32_t)idx, magic), x = idx - y * W2;
        const int oy = (int)__umulhi((uint32_t)oi, magic), ox = oi - oy * W2;
        const int dy = y - oy, dx = x - ox;
        const bool conflict = valid && n < ci && dy >= -2 && dy <= 2 && dx >= -2 && dx <= 2;
        const uint32_t clast = (uint32_t)__builtin_amdgcn_readlane((int)hot, mav - 1);
        const bool fresh = v >= 3u;
        const uint32_t e = fresh ? ((v << 16) | (uint32_t)np) : 0u;
        const unsigned long long cmC = __ballot(conflict);
        const unsigned long long cmH = __ballot(e > clast);
        int m = mav;
        if (cmC) { int g = (__ffsll((long long)cmC) - 1) >> 3; m = g < m ? g : m; BT_COUNT(3, g < mav); }
        if (cmH) { int g = ((__ffsll((long long)cmH) - 1) >> 3) + 1; BT_COUNT(4, g < m); m = g < m ? g : m; }
        BT_COUNT(0, 1); BT_COUNT(1, mav); BT_COUNT(2, m);

        // ---- commit candidates 0..m-1
        const bool act = ci < m;
        const unsigned long long visb = __ballot(act && v == 1u);
        const uint32_t grp = (uint32_t)(visb >> (ci * 8)) & 0xffu;
        // parent = lexicographically smallest visited neighbour = lowest neighbour slot with state 1
        const bool isp = act && v == 1u && (grp & ((1u << n) - 1u)) == 0u;
        if (isp) { ppar[idx] = np; order[npop + ci] = ((uint32_t)np << 16) | (uint32_t)idx; }
        if (act && n == 0) {
            kp[idx] = 1;
            if (grp == 0u) order[npop + ci] = ((uint32_t)idx << 16) | (uint32_t)idx;      // the seed: no visited neighbour, its own parent
        }
        npop += m;
        const bool ins = act && fresh;
        if (ins) kp[np] = 2;
        // drop the m popped entries from HOT
        {
            uint32_t sh = bt_bperm(hot, (lane + m) & 63);
            hot = lane + m < 64 ? sh : 0u;
            H -= m;
            if (H == 0) tailv = 0u;
        }
        // ---- new entries: HOT when they outrank the HOT tail (or the whole frontier is empty), else COLD
        unsigned long long insb = __ballot(ins);
        unsigned long long hotb = (H > 0) ? __ballot(ins && e > tailv) : (cold_any ? 0ull : insb);
        unsigned long long coldb = insb & ~hotb;
        while (hotb) {
            const int l = __ffsll((long long)hotb) - 1;
            hotb &= hotb - 1ull;
            const uint32_t el = (uint32_t)__builtin_amdgcn_readlane((int)e, l);
            if (H > 0 && !(el > tailv)) { coldb |= 1ull << l; continue; }   // the tail moved up meanwhile
            const uint32_t displaced = (uint32_t)__builtin_amdgcn_readlane((int)hot, 63);
            const uint32_t prev = bt_shr1(hot, 0xFFFFFFFFu);              // lane l <- l-1, lane 0 <- "infinity"
            hot = hot > el ? hot : (prev > el ? el : prev);               // branch-free sorted insertion
            if (H == 0) tailv = el;
            if (H < 64) H++;
            else {
                // HOT was full: its old tail falls into COLD
                const uint32_t tc = displaced >> 16;
                if (lane == 0) {
                    atomicOr(&L0[tc >> 6], 1ull << (tc & 63u));
                    atomicOr(&L1[tc >> 12], 1ull << ((tc >> 6) & 63u));
                }
                cold_any = true;
                tailv = (uint32_t)__builtin_amdgcn_readlane((int)hot, 63);
            }
        }
        if ((coldb >> lane) & 1ull) {
            const uint32_t cc = e >> 16;
            atomicOr(&L0[cc >> 6], 1ull << (cc & 63u));
            atomicOr(&L1[cc >> 12], 1ull << ((cc >> 6) & 63u));
        }
        if (coldb) cold_any = true;
    }
    if (lane == 0) order[ostride - 1] = (uint32_t)npop;
}

#ifdef VISTAF_DEBUG
void unwrap_batch_debug_dump()
{
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_batch_dbg), sizeof(h)) != hipSuccess) return;
    printf("[flood batch dbg] frame 0, all calls: steps %llu | candidates %llu | commits %llu | steps cut by (i) %llu (ii) %llu | refills %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
}
#endif

// k_unwrap_replay: integer wrap counts along the growth tree (shape_ftp.py:1060-1076) by replaying the pops in order.
// A pixel's count is its parent's plus the wrap of the phase step between them, and a parent always pops before
// its children, so walking the pop records front to back resolves everything in one sweep.  The counts live in
// LDS (int16 per padded pixel, RP_UNSET = not yet known); each wave takes every RP_NW-th chunk of 64 records,
// prefetches its phase values and spins on the parents' LDS entries, which earlier chunks (other waves) fill in.
constexpr int RP_NW = 16;
constexpr int16_t RP_UNSET = (int16_t)0x7fff;

__global__ __launch_bounds__(64 * RP_NW) void k_unwrap_replay(const float *__restrict__ wrapped_all, const uint32_t *__restrict__ order_all,
                                                              size_t ostride, const int32_t *__restrict__ ppar_all, size_t gstride,
                                                              int32_t *__restrict__ tree_all, float *__restrict__ unwrapped_all, int h, int w,
                                                              uint32_t magic, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int16_t *ks = (int16_t *)lds_raw;                                     // [EN8]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t b = blockIdx.x;
    const int P = h * w, W2 = w + 2, EN = (h + 2) * W2;
    const uint32_t *order = order_all + b * ostride;
    const float *wrapped = wrapped_all + b * (size_t)P;
    int32_t *tree = tree_all + b * (size_t)P;
    float *unwrapped = unwrapped_all + b * (size_t)P;
    const double twopi = 6.283185307179586476925286766559, pi_d = 3.14159265358979323846;
    {
        uint32_t fill = ((uint32_t)(uint16_t)RP_UNSET << 16) | (uint16_t)RP_UNSET;
        uint32_t *k32 = (uint32_t *)ks;
        for (int i = tid; i < ((EN + 7) & ~7) / 2; i += 64 * RP_NW) k32[i] = fill;
    }
    const int npop = (int)order[ostride - 1];
    __syncthreads();
    for (int c0 = wid * 64; c0 < npop; c0 += 64 * RP_NW) {
        const int i = c0 + lane;
        const bool valid = i < npop;
        const uint32_t rec = valid ? order[i] : 0u;
        const int idx = (int)(rec & 0xffffu), pp = (int)(rec >> 16);
        const int y = (int)__umulhi((uint32_t)idx, magic), x = idx - y * W2;
        const int py = (int)__umulhi((uint32_t)pp, magic), px = pp - py * W2;
        const int p = valid ? (y - 1) * w + (x - 1) : 0, par = valid ? (py - 1) * w + (px - 1) : 0;
        const float wp = wrapped[p], wq = wrapped[par];
        int v = 0;
        if (pp != idx) {
            double dd0 = (double)wp - (double)wq;
            double k = -rint(dd0 / twopi);
            double dd = dd0 + twopi * k;
            if (dd <= -pi_d) k += 1.0;
            else if (dd > pi_d) k -= 1.0;
            v = (int)k;
        }
        bool done = !valid;
        if (valid && pp == idx) { done = true; __hip_atomic_store(&ks[idx], (int16_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        while (__ballot(!done)) {
            if (!done) {
                int16_t kq = __hip_atomic_load(&ks[pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (kq != RP_UNSET) {
                    const int kf = (int)kq + v;
                    __hip_atomic_store(&ks[idx], (int16_t)kf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    done = true;
                }
            }
        }
    }
    // the two result planes in pixel order (coalesced; the walk above writes nothing to memory): unwrapped = wrapped + 2*pi*k, parent in
    // frame coordinates; NaN / -1 where the growth never arrived
    __syncthreads();
    const int32_t *ppar = ppar_all + b * gstride;
    for (int p = tid; p < P; p += 64 * RP_NW) {
        const int y = p / w, x = p - y * w;
        const int idx = (y + 1) * W2 + x + 1;
        const int16_t kq = ks[idx];
        float u = __uint_as_float(0x7fc00000u);
        int par = -1;
        if (kq != RP_UNSET) {
            u = (float)((double)wrapped[p] + twopi * (double)kq);
            const int pp = ppar[idx];
            const int py = (int)__umulhi((uint32_t)pp, magic), px = pp - py * W2;
            par = (py - 1) * w + (px - 1);
        }
        unwrapped[p] = u;
        tree[p] = par;
    }
}

bool unwrap_batch_supported(int h, int w)
{
    long EN = (long)(h + 2) * (w + 2);
    long lds = (((EN + 7) & ~7L)) * 2 + (BT_NW + 16) * 8 + 256 + 512;
    return EN <= 65533 && lds <= 160 * 1024;
}

void launch_unwrap_flood_batch(const uint16_t *rank16, const int32_t *seed, const uint32_t *inv, size_t inv_stride, int32_t *ppar, size_t gstride,
                               uint32_t *order, size_t ostride, int B, int h, int w, hipStream_t st, const int32_t *need)
{
    long EN = (long)(h + 2) * (w + 2);
    size_t lds = (size_t)(((EN + 7) & ~7L)) * 2 + (BT_NW + 16) * 8 + 256 + 512;
    static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_unwrap_flood_batch, 160 * 1024);
    const uint32_t magic = (uint32_t)(0x100000000ull / (unsigned)(w + 2)) + 1u;     // idx / (w + 2) == umulhi(idx, magic) for idx < 65536
    hipLaunchKernelGGL(k_unwrap_flood_batch, dim3(B), dim3(64), lds, st, rank16, seed, inv, inv_stride, ppar, gstride, order, ostride, h, w, magic, need);
}

// unwrapped = wrapped + 2*pi*k along the growth tree; NaN / parent -1 where the growth never arrived
void launch_unwrap_replay(const float *wrapped, const uint32_t *order, size_t ostride, const int32_t *ppar, size_t gstride, int32_t *tree,
                          float *unwrapped, int B, int h, int w, hipStream_t st, const int32_t *need)
{
    long EN = (long)(h + 2) * (w + 2);
    size_t lds = (size_t)(((EN + 7) & ~7L)) * 2;
    static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_unwrap_replay, 160 * 1024);
    const uint32_t magic = (uint32_t)(0x100000000ull / (unsigned)(w + 2)) + 1u;
    hipLaunchKernelGGL(k_unwrap_replay, dim3(B), dim3(64 * RP_NW), lds, st, wrapped, order, ostride, ppar, gstride, tree, unwrapped, h, w, magic, need);
}

}  // namespace vf
