// Fast path of unwrap_quality_guided (shape_ftp.py:1043-1080) for frames whose padded plane
// (h+2)*(w+2) has at most 65533 pixels.
//
// k_unwrap_rank   (1024 threads / frame) replaces each masked pixel's float quality by its RANK in the
//                 frame's total order (q ascending, ties: larger pixel index first, so that the larger
//                 rank is exactly the reference heap's higher priority "-q, then smaller (y, x)").
//                 Stable LSD radix sort, 11-bit digits (3 passes), wave-contiguous chunks, 4 tiles of loads in flight.  The rank codes are
//                 written into a plane padded by one pixel of zeros on every side.
// k_unwrap_flood_ranked (one wavefront / frame) holds the whole padded frame in LDS as one uint16 per
//                 pixel (0 outside mask / border, 1 visited, 2 in frontier, >= 3 untouched with
//                 rank = v - 3).  The sequential growth loop touches no global memory except the
//                 parent store, needs no bounds checks or divisions (the border is "outside"), reads the
//                 8 neighbours with 8 lanes, and finds the frontier maximum with a packed b128 scan of
//                 uint16 keys plus one DPP wave reduction.  Keys are unique: no tie handling.
#include <cstdlib>
#include <cstring>
#include "kernels.hpp"
#include "select.hpp"

namespace vf {

bool unwrap_hot_supported(int h, int w);
bool unwrap_batch_supported(int h, int w);
void launch_unwrap_flood_batch(const uint16_t *rank16, const int32_t *seed, const uint32_t *inv, size_t inv_stride, int32_t *ppar, size_t gstride,
                               uint32_t *order, size_t ostride, int B, int h, int w, hipStream_t st, const int32_t *need);
void launch_unwrap_flood_hot(const uint16_t *rank16, const int32_t *seed, const uint32_t *inv, size_t inv_stride, int32_t *ppar, size_t gstride,
                             int32_t *status, int B, int h, int w, hipStream_t st, const int32_t *need);

constexpr int RK_T = 1024;

// ---- DPP wave reductions (gfx9 row / bcast controls) -------------------------------------------------
__device__ inline uint32_t dpp_max_u32(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false); v = t > v ? t : v;    // quad_perm [1,0,3,2]
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false); v = t > v ? t : v;    // quad_perm [2,3,0,1]
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false); v = t > v ? t : v;   // row_half_mirror
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false); v = t > v ? t : v;   // row_mirror
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;   // row_bcast15 -> rows 1,3
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;   // row_bcast31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// maximum over lanes 0..7 (valid in lane 0..7), returned uniform
__device__ inline uint32_t dpp_max8_u32(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false); v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// ---- ranks -------------------------------------------------------------------------------------------
// rank plane layout: [(h+2) x (w+2)] uint16 (frame stride padded to 8 elements), border = 0.
// Sort structure: each of the 16 waves owns a contiguous range of the element array and walks it in
// 64-element tiles (coalesced, L1-bypassing loads of data other waves wrote in the previous pass).
// Counting uses a per-wave 2048-bin LDS histogram (16 x 8 KB); the stable scatter ranks a lane among the lanes of its
// tile that share its digit with eleven ballots (peer mask) + mbcnt.
// Every exchange through global memory in this kernel is between waves of ONE workgroup (one frame): workgroup scope is all the
// ordering it needs.  (Agent scope would write back and invalidate the XCD's whole L2 at every fence: buffer_wbl2 / buffer_inv sc1.)
__device__ inline uint32_t ld_u32c(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline unsigned long long ld_u64c(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

constexpr int RK_BITS = 11, RK_NB = 1 << RK_BITS;   // 3 passes of 11 bits over the 32-bit keys
constexpr int RK_U = 4;        // 64-element tiles in flight per wave (independent loads issued together)

// CodeT = uint16_t: frames of up to 65533 padded pixels (the LDS-resident floods); uint32_t: larger frames (k_unwrap_big.hip), which also
// get the number of masked pixels in n_out
template <typename CodeT>
__device__ __attribute__((always_inline)) inline void unwrap_rank_body(const float *__restrict__ quality_all, const uint8_t *__restrict__ mask_all,
                                                                       unsigned long long *A_all, unsigned long long *B_all, size_t gstride,
                                                                       CodeT *__restrict__ rank_all, int32_t *__restrict__ seed_out,
                                                                       int32_t *__restrict__ n_out, int h, int w)
{
    // sort records: key << 32 | padded pixel index (one 8-byte scattered store per element and pass)
    extern __shared__ uint32_t rk_lds[];
    uint32_t (*whist)[RK_NB] = (uint32_t (*)[RK_NB])rk_lds;    // [16 waves][RK_NB digits] counts, then exclusive offsets (128 KB)
    __shared__ uint32_t wcount[16];
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int P = h * w, W2 = w + 2, EN = (h + 2) * W2;
    const float *q = quality_all + b * (size_t)P;
    const uint8_t *m = mask_all + b * (size_t)P;
    unsigned long long *rA = A_all + b * gstride, *rB = B_all + b * gstride;
    uint32_t *inv = (uint32_t *)rA;             // three passes leave the records in rB: rA is free for the sorted pixel indices
    CodeT *rk = rank_all + b * (size_t)((EN + 7) & ~7);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (int i = tid; i < EN; i += RK_T) rk[i] = 0;
    // 1. compact masked pixels in DESCENDING pixel order; wave `wid` owns reversed positions [r0, r1)
    const int Lw = (((P + 15) / 16) + 63) & ~63;
    const int r0 = min(P, wid * Lw), r1 = min(P, r0 + Lw);
    uint32_t c = 0;
    for (int rb = r0; rb < r1; rb += 64 * RK_U) {
        uint8_t mm[RK_U];
#pragma unroll
        for (int u = 0; u < RK_U; u++) { int r = rb + u * 64 + lane; mm[u] = r < r1 ? m[P - 1 - r] : (uint8_t)0; }
#pragma unroll
        for (int u = 0; u < RK_U; u++) c += (uint32_t)__popcll(__ballot(mm[u] != 0));
    }
    if (lane == 0) wcount[wid] = c;
    __syncthreads();
    uint32_t off = 0, n = 0;
    for (int i = 0; i < 16; i++) { uint32_t x = wcount[i]; if (i < wid) off += x; n += x; }
    for (int rb = r0; rb < r1; rb += 64 * RK_U) {
        uint8_t mm[RK_U];
        float qq[RK_U];
#pragma unroll
        for (int u = 0; u < RK_U; u++) {
            int r = rb + u * 64 + lane;
            bool in = r < r1;
            mm[u] = in ? m[P - 1 - r] : (uint8_t)0;
            qq[u] = in ? q[P - 1 - r] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < RK_U; u++) {
            bool in = mm[u] != 0;
            unsigned long long bm = __ballot(in);
            if (in) {
                int p = P - 1 - (rb + u * 64 + lane);
                int y = p / w, x = p - y * w;
                uint32_t o = off + (uint32_t)__popcll(bm & lt_mask);
                rA[o] = ((unsigned long long)f2key(qq[u]) << 32) | (uint32_t)((y + 1) * W2 + x + 1);
            }
            off += (uint32_t)__popcll(bm);
        }
    }
    if (tid == 0 && n == 0) seed_out[b] = -1;
    if (tid == 0 && n_out) n_out[b] = (int32_t)n;
    __threadfence_block();
    __syncthreads();
    if (n == 0) return;

    // 2. stable LSD radix sort, 3 passes of 11 bits; wave `wid` owns elements [e0, e1)
    const int Mw = ((((int)n + 15) / 16) + 63) & ~63;
    const int e0 = min((int)n, wid * Mw), e1 = min((int)n, e0 + Mw);
    unsigned long long *rs = rA, *rd = rB;
    for (int pass = 0; pass < 3; pass++) {
        const int shift = pass * RK_BITS;
        for (int i = lane; i < RK_NB; i += 64) whist[wid][i] = 0;
        for (int eb = e0; eb < e1; eb += 64 * RK_U) {
            uint32_t kk[RK_U];
#pragma unroll
            for (int u = 0; u < RK_U; u++) { int e = eb + u * 64 + lane; kk[u] = e < e1 ? ld_u32c((const uint32_t *)&rs[e] + 1) : 0u; }
#pragma unroll
            for (int u = 0; u < RK_U; u++)
                if (eb + u * 64 + lane < e1) atomicAdd(&whist[wid][(kk[u] >> shift) & (RK_NB - 1)], 1u);
        }
        __syncthreads();
        // exclusive offsets in (digit-major, wave-minor) order: thread t owns digits 2t and 2t + 1 of all 16 waves
        {
            uint32_t mine = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) mine += whist[k & 15][2 * tid + (k >> 4)];
            uint32_t incl = wave_scan_add(mine);
            if (lane == 63) wcount[wid] = incl;
            __syncthreads();
            uint32_t run = incl - mine;
            for (int i = 0; i < wid; i++) run += wcount[i];
#pragma unroll
            for (int k = 0; k < 32; k++) {
                uint32_t v = whist[k & 15][2 * tid + (k >> 4)];
                whist[k & 15][2 * tid + (k >> 4)] = run;
                run += v;
            }
        }
        __syncthreads();
        for (int eb = e0; eb < e1; eb += 64 * RK_U) {
            unsigned long long rr[RK_U];
#pragma unroll
            for (int u = 0; u < RK_U; u++) {
                int e = eb + u * 64 + lane;
                rr[u] = e < e1 ? ld_u64c(&rs[e]) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < RK_U; u++) {
                const bool ok = eb + u * 64 + lane < e1;
                const uint32_t dgt = (uint32_t)(rr[u] >> (32 + shift)) & (RK_NB - 1);
                unsigned long long peers = __ballot(ok);
#pragma unroll
                for (int bit = 0; bit < RK_BITS; bit++) {
                    unsigned long long bm = __ballot(ok && ((dgt >> bit) & 1u));
                    peers &= ((dgt >> bit) & 1u) ? bm : ~bm;
                }
                const uint32_t base = whist[wid][dgt];
                const uint32_t rnk = (uint32_t)__popcll(peers & lt_mask);
                if (ok) rd[base + rnk] = rr[u];
                __builtin_amdgcn_wave_barrier();
                if (ok && rnk == 0) whist[wid][dgt] = base + (uint32_t)__popcll(peers);
                __builtin_amdgcn_wave_barrier();
            }
        }
        __threadfence_block();
        __syncthreads();
        unsigned long long *t = rs; rs = rd; rd = t;
    }
    // 3. rank = sorted position (ascending priority); uint16 code = rank + 3; inv[rank] = padded pixel index
    for (int e = tid; e < (int)n; e += RK_T) {
        uint32_t ix = (uint32_t)ld_u64c(&rs[e]);
        rk[ix] = (CodeT)(e + 3);
        inv[e] = ix;
        if (e == (int)n - 1) seed_out[b] = (int32_t)ix;
    }
}
__global__ __launch_bounds__(RK_T) void k_unwrap_rank(const float *__restrict__ quality_all, const uint8_t *__restrict__ mask_all,
                                                      unsigned long long *A_all, unsigned long long *B_all, size_t gstride,
                                                      uint16_t *__restrict__ rank_all, int32_t *__restrict__ seed_out, int h, int w, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    unwrap_rank_body<uint16_t>(quality_all, mask_all, A_all, B_all, gstride, rank_all, seed_out, nullptr, h, w);
}
__global__ __launch_bounds__(RK_T) void k_unwrap_rank32(const float *__restrict__ quality_all, const uint8_t *__restrict__ mask_all,
                                                        unsigned long long *A_all, unsigned long long *B_all, size_t gstride,
                                                        uint32_t *__restrict__ rank_all, int32_t *__restrict__ seed_out, int32_t *__restrict__ n_out,
                                                        int h, int w, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    unwrap_rank_body<uint32_t>(quality_all, mask_all, A_all, B_all, gstride, rank_all, seed_out, n_out, h, w);
}
// ranks of frames too large for uint16 codes: 32-bit codes in a padded plane, sorted pixel indices in A (uint32, stride 2 * gstride)
void launch_unwrap_rank32(const float *quality, const uint8_t *mask, uint32_t *gA, uint32_t *gB, size_t gstride, uint32_t *rank32, int32_t *seed,
                          int32_t *n_out, int B, int h, int w, hipStream_t st, const int32_t *need)
{
    static DynLdsOnce rank_once;
    ensure_dyn_lds(rank_once, (const void *)k_unwrap_rank32, 16 * RK_NB * (int)sizeof(uint32_t));
    hipLaunchKernelGGL(k_unwrap_rank32, dim3(B), dim3(RK_T), (size_t)16 * RK_NB * sizeof(uint32_t), st, quality, mask, (unsigned long long *)gA,
                       (unsigned long long *)gB, gstride, rank32, seed, n_out, h, w, need);
}

// ---- growth --------------------------------------------------------------------------------------------
// ppar[padded pixel] = padded index of its parent (own index for the seed), -1 where never reached
__global__ __launch_bounds__(64) void k_unwrap_flood_ranked(const uint16_t *__restrict__ rank_all, const int32_t *__restrict__ seed_in,
                                                            int32_t *__restrict__ ppar_all, size_t gstride, int cap, int32_t *status,
                                                            int h, int w, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    const int W2 = w + 2, EN = (h + 2) * W2;
    const int EN8 = (EN + 7) & ~7;
    uint16_t *kp = (uint16_t *)lds_raw;                // [EN8] pixel state / rank code
    uint32_t *fe = (uint32_t *)(kp + EN8);             // [cap] frontier entries: rank code << 16 | padded pixel index (0 = empty)
    int32_t *ppar = ppar_all + b * gstride;
    const uint16_t *rk = rank_all + b * (size_t)((EN + 7) & ~7);   // frame stride padded to 16 bytes

    {
        const uint4 *src = (const uint4 *)rk;
        uint4 *dst = (uint4 *)kp;
        int nv = EN >> 3;
        for (int i = lane; i < nv; i += 64) dst[i] = src[i];
        for (int p = (nv << 3) + lane; p < EN; p += 64) kp[p] = rk[p];
    }
    {
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *dst = (uint4 *)fe;
        for (int i = lane; i < (cap >> 2); i += 64) dst[i] = z;
    }
    for (int p = lane; p < EN; p += 64) ppar[p] = -1;
    __syncthreads();
    int cur = seed_in[b];
    if (cur < 0) return;                               // empty mask (shape_ftp.py:1047-1048)
    // neighbour offsets of lanes 0..7 in lexicographic (dy, dx) order
    int doff = 0;
    {
        int l = (lane & 7) < 4 ? (lane & 7) : (lane & 7) + 1;
        doff = (l / 3 - 1) * W2 + (l % 3 - 1);
    }
    int F = 0;
    bool first = true, overflow = false;
    if (lane == 0) kp[cur] = 1;
    const uint4 *fe4 = (const uint4 *)fe;

    for (;;) {
        // all LDS reads of the step are issued back to back (one round trip): the 8 neighbours of `cur`,
        // the frontier entries, and the last entry (it fills the hole if an old entry is popped)
        uint32_t v = 0;
        int np = cur + doff;
        if (lane < 8) v = kp[np];
        uint32_t lastv = F > 0 ? fe[F - 1] : 0u;
        uint32_t best = 0, bv0 = 0, bv1 = 0, bv2 = 0, bv3 = 0;
        int bbase = 0;
        // four independent b128 loads per trip are issued before the first use (cap is a multiple of 1024 and
        // slots >= F are zero, so over-reading a trip is harmless)
        for (int base = 0; base < F; base += 1024) {
            const uint4 *src = fe4 + (base >> 2) + lane;
            uint4 k0 = src[0], k1 = src[64], k2 = src[128], k3 = src[192];
#define VF_SCAN4(kv, off)                                                                                      \
            {                                                                                                  \
                uint32_t m01 = kv.x > kv.y ? kv.x : kv.y, m23 = kv.z > kv.w ? kv.z : kv.w;                       \
                uint32_t m = m01 > m23 ? m01 : m23;                                                            \
                if (m > best) { best = m; bbase = base + (off); bv0 = kv.x; bv1 = kv.y; bv2 = kv.z; bv3 = kv.w; } \
            }
            VF_SCAN4(k0, 0) VF_SCAN4(k1, 256) VF_SCAN4(k2, 512) VF_SCAN4(k3, 768)
#undef VF_SCAN4
        }
        unsigned long long vis = __ballot(v == 1);
        bool fresh = v >= 3;
        unsigned long long fb = __ballot(fresh);
        int par = cur;
        if (!first) par = cur + __builtin_amdgcn_readlane(doff, __ffsll((long long)vis) - 1);
        first = false;
        if (lane == 0) ppar[cur] = par;
        uint32_t mq = dpp_max_u32(best);               // rank code in the high half, pixel index in the low half
        uint32_t nk = fresh ? v : 0u;
        uint32_t nm = dpp_max8_u32(nk);
        if ((mq | nm) == 0) break;                     // frontier exhausted
        int next;
        unsigned long long app = fb;                   // lanes whose neighbour is appended to the frontier
        if (nm > (mq >> 16)) {
            int wl = __ffsll((long long)__ballot(fresh && nk == nm)) - 1;
            next = cur + __builtin_amdgcn_readlane(doff, wl);
            app &= ~(1ull << wl);
        } else {
            next = (int)(mq & 0xffffu);
            int wl = __ffsll((long long)__ballot(best == mq)) - 1;
            int sub = bv0 == mq ? 0 : bv1 == mq ? 1 : bv2 == mq ? 2 : 3;
            int slot = __builtin_amdgcn_readlane(bbase + (lane << 2) + sub, wl);
            int last = F - 1;
            if (lane == 0) { fe[slot] = lastv; fe[last] = 0; }     // slot == last: the second store wins
            F = last;
        }
        int napp = (int)__popcll(app);
        if (F + napp > cap) { overflow = true; break; }
        if ((app >> lane) & 1ull) {
            int pos = F + (int)__popcll(app & ((1ull << lane) - 1ull));
            fe[pos] = (nk << 16) | (uint32_t)np;
            kp[np] = 2;
        }
        F += napp;
        cur = next;
        if (lane == 0) kp[cur] = 1;
    }
    if (overflow && lane == 0) status[b] = 2;
}

static int ranked_cap(int EN)
{
    long plane = (long)(((EN + 7) & ~7)) * 2;
    long avail = 160 * 1024 - plane;
    long cap = (avail / 4) & ~1023L;
    return (int)cap;
}

bool unwrap_ranked_supported(int h, int w)
{
    long EN = (long)(h + 2) * (w + 2);
    return EN <= 65533 && ranked_cap((int)EN) >= 2048;
}

// g0..g3: uint32 planes of gstride elements per frame (sort ping-pong); ppar: int32 plane of gstride elements
// returns true when the growth kernel also left its pop records in g2 (stride 2 * gstride per frame) for launch_unwrap_replay
bool launch_unwrap_ranked(const float *quality, const uint8_t *mask, uint32_t *g0, uint32_t *g1, uint32_t *g2, uint32_t *g3,
                          int32_t *ppar, size_t gstride, uint16_t *rank16, int32_t *seed, int32_t *status, int B, int h, int w,
                          hipStream_t st, hipEvent_t ev_flood, int flood_tier, const int32_t *need)
{
    int EN = (h + 2) * (w + 2);
    int cap = ranked_cap(EN);
    // g0|g1 and g2|g3 are contiguous (k_unwrap.hip): two planes of 8-byte sort records; the sorted pixel indices go to g2
    (void)g1; (void)g3;
    static DynLdsOnce rank_once;
    ensure_dyn_lds(rank_once, (const void *)k_unwrap_rank, 16 * RK_NB * (int)sizeof(uint32_t));   // + 64 B static
    hipLaunchKernelGGL(k_unwrap_rank, dim3(B), dim3(RK_T), (size_t)16 * RK_NB * sizeof(uint32_t), st, quality, mask, (unsigned long long *)g0,
                       (unsigned long long *)g2, gstride, rank16, seed, h, w, need);
    if (ev_flood) hipEventRecord(ev_flood, st);
    // growth loop (Tiers::flood): 2 = "batch" (default: 8 pops per step, k_unwrap_batch.hip), 1 = "hot" (one pop per step, sorted
    // register list + rank bitmap), 0 = "scan" (frontier array scan)
    const int use_hot = flood_tier;
    if (use_hot == 2 && unwrap_batch_supported(h, w)) {
        // sorted pixel indices: g0 (stride 2 * gstride); the sort records in g2|g3 are dead once the ranks are out: the growth
        // kernel logs its pops there
        launch_unwrap_flood_batch(rank16, seed, g0, 2 * gstride, ppar, gstride, g2, 2 * gstride, B, h, w, st, need);
        return true;
    }
    if (use_hot && unwrap_hot_supported(h, w)) {
        launch_unwrap_flood_hot(rank16, seed, g0, 2 * gstride, ppar, gstride, status, B, h, w, st, need);
        return false;
    }
    static DynLdsOnce lds_once;
    ensure_dyn_lds(lds_once, (const void *)k_unwrap_flood_ranked, 160 * 1024);
    size_t lds = (size_t)((EN + 7) & ~7) * 2 + (size_t)cap * 4;
    hipLaunchKernelGGL(k_unwrap_flood_ranked, dim3(B), dim3(64), lds, st, rank16, seed, ppar, gstride, cap, status, h, w, need);
    return false;
}

}  // namespace vf
