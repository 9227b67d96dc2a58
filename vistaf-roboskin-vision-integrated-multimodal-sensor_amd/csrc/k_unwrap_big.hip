// k_unwrap_flood_big: the growth loop of unwrap_quality_guided (shape_ftp.py:1043-1080) for frames whose padded plane does not fit the LDS
// (more than 65533 padded pixels: the native 1182 x 1182 crops).  Same contract as the LDS-resident floods (k_unwrap_batch.hip): the total
// order of the reference's heap, (-q, y, x), is turned into RANKS up front (k_unwrap_rank32), the frontier is the set of ranks whose bit is
// set, a pop is "highest set bit", the parent of a popped pixel is its lexicographically smallest visited neighbour.
//
//   * the pixel state / rank-code plane (uint32 per padded pixel: 0 outside the mask, 1 visited, 2 frontier, >= 3 rank + 3) stays in
//     GLOBAL memory -- 5.6 MB per native frame;
//   * the priority queue is a three-level bitmap over the ranks in LDS: 19 200 words = 1.23 M ranks (the ROI disc of a native crop holds
//     1.09 M pixels), one summary bit per word on the two levels above.  Push = three LDS ORs, pop = three dependent LDS reads;
//   * one wave per frame, up to 8 pops per step as in k_unwrap_flood_batch: the candidates are the 8 highest set bits of the up to 8 highest
//     non-empty words under the top summary word (lanes 0..7 read their words together), their pixels come from the sorted-index array of the rank kernel, lane = candidate * 8 + neighbour
//     loads the 8 neighbour codes in ONE global round trip, and the longest prefix that provably pops in that order with the same
//     neighbour states as the one-at-a-time loop is committed:
//       (i)  a candidate within Chebyshev distance 2 of an earlier candidate ends the prefix before it (it could see the earlier one as a
//            visited neighbour or share a fresh neighbour with it);
//       (ii) a fresh neighbour that outranks a later candidate has to pop before it: the prefix ends before that candidate.
//   A step costs two dependent global round trips (sorted index -> pixel, pixel -> neighbour codes) instead of a scan of the whole
//   frontier array per pop (k_unwrap_flood<false>, which stays the fallback for masks of more than 1.23 M pixels).
#include <cstdio>
#include "kernels.hpp"

namespace vf {

namespace {
constexpr int BG_NW0 = 19200;                        // L0 words: 150 KB
constexpr int BG_K = 8;                              // candidates per step

__device__ inline uint32_t bg_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void bg_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int bg_top(unsigned long long v) { return 63 - __clzll((long long)v); }
__device__ inline unsigned long long bg_lane64(unsigned long long v, int l)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
#ifdef VISTAF_DEBUG
__device__ unsigned long long g_big_dbg[8];        // frame 0: steps, candidates, commits, steps cut by rule (i) / (ii) / (iii)
#define BG_COUNT(i, v) do { if (b == 0 && lane == 0) g_big_dbg[i] += (unsigned long long)(v); } while (0)
#else
#define BG_COUNT(i, v) do { } while (0)
#endif
}  // namespace

__global__ __launch_bounds__(64) void k_unwrap_flood_big(uint32_t *__restrict__ code_all, const int32_t *__restrict__ seed_in,
                                                         const int32_t *__restrict__ n_in, const uint32_t *__restrict__ inv_all, size_t inv_stride,
                                                         int32_t *__restrict__ ppar_all, size_t gstride, int32_t *__restrict__ need_generic, int max_ranks,
                                                         int nw0, int h, int w, uint32_t magic, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // nw0 words of L0 (sized by the host for the frame: every mask pixel can be a rank), then the two summary levels
    const int nw1 = (nw0 + 63) >> 6, nw2 = (nw1 + 63) >> 6;
    unsigned long long *L0 = (unsigned long long *)lds_raw;
    unsigned long long *L1 = L0 + nw0;
    unsigned long long *L2 = L1 + nw1;
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    const int W2 = w + 2, EN = (h + 2) * W2;
    uint32_t *code = code_all + b * (size_t)((EN + 7) & ~7);
    const uint32_t *inv = inv_all + b * inv_stride;
    int32_t *ppar = ppar_all + b * gstride;
    const int n = __builtin_amdgcn_readfirstlane(n_in[b]);
    const int seed = __builtin_amdgcn_readfirstlane(seed_in[b]);
    for (int p = lane; p < EN; p += 64) ppar[p] = -1;
    if (lane == 0) need_generic[b] = 0;
    if (seed < 0 || n <= 0) return;                                      // empty mask (shape_ftp.py:1047-1048)
    if (n > max_ranks) { if (lane == 0) need_generic[b] = 1; return; }       // mask larger than the bitmap: the generic kernel takes the frame
    for (int i = lane; i < nw0 + nw1 + nw2; i += 64) L0[i] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    // the seed (rank n - 1) is the first frontier entry
    if (lane == 0) {
        const int r = n - 1;
        L0[r >> 6] = 1ull << (r & 63);
        L1[r >> 12] = 1ull << ((r >> 6) & 63);
        L2[r >> 18] = 1ull << ((r >> 12) & 63);
        bg_st(code + seed, 2u);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    const int ci = lane >> 3, nb = lane & 7;                              // candidate slot, neighbour slot
    int doff;
    {
        const int l = nb < 4 ? nb : nb + 1;                              // the 3 x 3 block without its centre, lexicographic (dy, dx)
        doff = (l / 3 - 1) * W2 + (l % 3 - 1);
    }
    // Candidates of a step: the top set bit of up to 8 consecutive non-empty L0 words under the top non-empty L1 word (three dependent LDS
    // rounds), and the read of their pixels from the sorted-index array.  It runs at the END of a step, before the fence that drains the
    // step's stores: the bitmap is final by then, and the index read overlaps the drain.
    int C = 0, myw0 = -1, mywd = -1, myrank = -1;        // mywd: the L0 word of this lane's candidate
    uint32_t mypix = 0;
    auto search = [&]() {
        C = 0; myw0 = -1; mywd = -1; myrank = -1; mypix = 0;
        const unsigned long long v2 = lane < nw2 ? L2[lane] : 0ull;
        const unsigned long long nz2 = __ballot(v2 != 0ull);
        if (nz2 == 0ull) return;                                         // frontier exhausted: C = 0
        const int t2 = 63 - __clzll((long long)nz2);
        const unsigned long long top2 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v2 >> 32), t2) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)v2, t2);
        const int w1 = t2 * 64 + bg_top(top2);
        unsigned long long v1 = L1[w1];                                  // uniform, non-zero
        // lane k < 8 takes the k-th highest set bit of v1
        {
            unsigned long long t = v1;
            for (int k = 0; k < BG_K; k++) {
                if (t == 0ull) break;
                const int bt = bg_top(t);
                if (lane == k) myw0 = w1 * 64 + bt;
                t &= ~(1ull << bt);
            }
        }
        const unsigned long long v0 = (lane < BG_K && myw0 >= 0) ? L0[myw0] : 0ull;
        const int have = (int)__popcll(__ballot(lane < BG_K && myw0 >= 0));
        // the 8 highest frontier ranks: the words in descending order, the bits of a word in descending order (in smooth quality maps the top
        // of the frontier is a run of neighbouring ranks, i.e. several bits of ONE word)
        {
            int cnt = 0;
            for (int j = 0; j < have && cnt < BG_K; j++) {
                unsigned long long t = bg_lane64(v0, j);
                const int w0j = __builtin_amdgcn_readlane(myw0, j);
                while (t != 0ull && cnt < BG_K) {
                    const int bt = bg_top(t);
                    if (lane == cnt) { myrank = w0j * 64 + bt; mywd = w0j; }
                    t &= ~(1ull << bt);
                    cnt++;
                }
            }
            C = cnt;
        }
        BG_COUNT(6, have);
        if (lane < C) mypix = inv[myrank];
    };
    search();
    // every step commits at least candidate 0, so n steps are an upper bound (a guard, not a schedule)
    for (int step = 0; step < n && C > 0; step++) {
        // ---- one global round trip: the 8 neighbour codes of every candidate
        const int cpix = (int)(uint32_t)__builtin_amdgcn_ds_bpermute(ci << 2, (int)mypix);
        const bool live = ci < C;
        const int np = cpix + doff;
        uint32_t cd = 0;
        if (live) cd = bg_ld(code + np);
        // ---- the prefix that pops in this order
        // (i) candidate j is blocked by an earlier candidate within Chebyshev distance 2: lane = j * 8 + i tests the pair (i < j)
        int L;
        {
            const int pj = cpix, pi = (int)(uint32_t)__builtin_amdgcn_ds_bpermute(nb << 2, (int)mypix);
            // row = index / W2 as a multiply-high (exact while EN * W2 < 2^32; magic == 0: plain division)
            const int yj = magic ? (int)__umulhi((uint32_t)pj, magic) : pj / W2, yi = magic ? (int)__umulhi((uint32_t)pi, magic) : pi / W2;
            const int xj = pj - yj * W2, xi = pi - yi * W2;
            const bool close = live && nb < ci && abs(yj - yi) <= 2 && abs(xj - xi) <= 2;
            const unsigned long long cl = __ballot(close);
            L = C;
            if (cl) L = min(L, (int)((__ffsll((long long)cl) - 1) >> 3));
            BG_COUNT(3, L < C);
        }
        // (ii) a fresh neighbour of candidate i that outranks a LATER candidate k has to pop before k (and whatever it pushes in turn could
        // touch k's neighbourhood): the prefix ends before the first such k.  Candidate ranks descend, so k0 = the first k > i below it.
        {
            const bool fresh = live && cd >= 3u;
            const int f = (int)(cd - 3u);
            int k0 = BG_K;
#pragma unroll
            for (int k = BG_K - 1; k >= 1; k--) {
                const int rk_k = __builtin_amdgcn_readlane(myrank, k);       // -1 beyond the list
                if (fresh && k > ci && k < C && f > rk_k) k0 = k;
            }
#pragma unroll
            for (int k = 1; k < BG_K; k++) {
                if (k >= L) break;
                if (__ballot(k0 <= k)) { L = k; BG_COUNT(4, 1); break; }
            }
        }
        BG_COUNT(0, 1); BG_COUNT(1, C); BG_COUNT(2, L);
        // ---- commit candidates 0..L-1
        const bool com = ci < L;
        const unsigned long long vis = __ballot(com && cd == 1u);
        const bool fr = com && cd >= 3u;
        // parent: the first visited neighbour in lexicographic order; none: the seed, its own parent (shape_ftp.py:1051)
        {
            const unsigned int vis8 = (unsigned int)(vis >> (ci << 3)) & 0xffu;
            if (com && nb == 0) {
                int par = cpix;
                if (vis8) {
                    const int l0 = __ffs((int)vis8) - 1;
                    const int l = l0 < 4 ? l0 : l0 + 1;
                    par = cpix + (l / 3 - 1) * W2 + (l % 3 - 1);
                }
                ppar[cpix] = par;
                bg_st(code + cpix, 1u);
            }
        }
        // clear the committed candidates' bits (several may share an L0 word: the atomic that empties it clears the summaries)
        if (lane < L) {
            const unsigned long long bit = 1ull << (myrank & 63);
            const unsigned long long old = atomicAnd(&L0[mywd], ~bit);
            if ((old & ~bit) == 0ull) {
                const unsigned long long b1 = 1ull << (mywd & 63);
                const unsigned long long o1 = atomicAnd(&L1[mywd >> 6], ~b1);
                if ((o1 & ~b1) == 0ull) atomicAnd(&L2[mywd >> 12], ~(1ull << ((mywd >> 6) & 63)));
            }
        }
        __builtin_amdgcn_wave_barrier();
        // fresh neighbours enter the frontier
        if (fr) {
            const int r = (int)(cd - 3u);
            bg_st(code + np, 2u);
            atomicOr(&L0[r >> 6], 1ull << (r & 63));
            atomicOr(&L1[r >> 12], 1ull << ((r >> 6) & 63));
            atomicOr(&L2[r >> 18], 1ull << ((r >> 12) & 63));
        }
        __builtin_amdgcn_wave_barrier();
        search();                                                        // next step's candidates; its index read overlaps the drain below
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

#ifdef VISTAF_DEBUG
void unwrap_big_debug_dump()
{
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_big_dbg), sizeof(h)) != hipSuccess) return;
    printf("[flood big dbg] frame 0, all calls: steps %llu | candidates %llu (words found %llu) | commits %llu | steps cut by (i) %llu (ii) %llu (iii) %llu\n", h[0], h[1],
           h[6], h[2], h[3], h[4], h[5]);
}
#endif

bool unwrap_big_supported(int h, int w)
{
    const long EN = (long)(h + 2) * (w + 2);
    return EN > 65533 && EN < (1L << 26);
}

// force_generic (test hook flood_tier = 3): hand every frame back, which exercises the per-frame fallback plumbing at small sizes
void launch_unwrap_flood_big(uint32_t *code, const int32_t *seed, const int32_t *n, const uint32_t *inv, size_t inv_stride, int32_t *ppar,
                             size_t gstride, int32_t *need_generic, bool force_generic, int B, int h, int w, hipStream_t st, const int32_t *need)
{
    // the bitmap is sized for the frame (at most one rank per pixel), up to the 150 KB of BG_NW0 words: mid-size frames leave LDS to others
    const long P = (long)h * w;
    const unsigned long long EN = (unsigned long long)(h + 2) * (w + 2), W2 = (unsigned long long)w + 2;
    const uint32_t magic = EN * W2 < 0x100000000ull ? (uint32_t)(0x100000000ull / W2) + 1u : 0u;
    const int nw0 = (int)std::min<long>(BG_NW0, (P + 63) / 64);
    const int nw1 = (nw0 + 63) / 64, nw2 = (nw1 + 63) / 64;
    const size_t lds = (size_t)(nw0 + nw1 + nw2) * 8 + 64;
    static DynLdsOnce lds_once;
    ensure_dyn_lds(lds_once, (const void *)k_unwrap_flood_big, 160 * 1024);
    hipLaunchKernelGGL(k_unwrap_flood_big, dim3(B), dim3(64), lds, st, code, seed, n, inv, inv_stride, ppar, gstride, need_generic,
                       force_generic ? 0 : nw0 * 64, nw0, h, w, magic, need);
}

}  // namespace vf
