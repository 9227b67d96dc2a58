// unwrap_quality_guided on the GPU (shape_ftp.py:1043-1080).
//
// The reference grows a region from the best-quality pixel with a max-heap of tuples
// (-q, y, x, py, px).  That is Prim-style growth with a total order: the next pixel is the frontier
// pixel with the largest (q, then smallest (y, x)), and its parent is the lexicographically smallest
// (py, px) among its neighbours visited so far.  The growth order is inherently sequential, so:
//   k_unwrap_flood  one wavefront per frame replays exactly that order (frontier keys in LDS, 64-lane
//                   arg-max per step, the 8 neighbours examined by 8 lanes) and records only the
//                   spanning tree (parent index per pixel);
//   k_unwrap_inc / k_unwrap_jump / k_unwrap_apply  (parallel) turn the tree into integer wrap counts
//                   by pointer jumping (integer sums are associative, so any evaluation order gives
//                   the tree's exact counts) and write u = w + 2*pi*k.
// In exact arithmetic u[p] - w[p] is the same multiple of 2*pi as in the reference's
// u[p] = u[parent] + angle(exp(1j*(w[p]-w[parent]))) chain; only float32 rounding along the chain differs.
#include "kernels.hpp"

namespace vf {

__device__ inline uint8_t ld_st(const uint8_t *st, int i, bool lds)
{
    if (lds) return st[i];
    return __hip_atomic_load(st + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool LS>
__global__ __launch_bounds__(64) void k_unwrap_flood(const float *__restrict__ quality_all, const uint8_t *__restrict__ mask_all,
                                                     int32_t *__restrict__ parent_all, uint8_t *gst, uint32_t *gfq, uint32_t *gfi,
                                                     int cap, int32_t *status, int h, int w, const int32_t *__restrict__ only, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    extern __shared__ unsigned char lds_raw[];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    if (only && !only[b]) return;                    // big-frame path: only the frames the bitmap flood handed back
    const int P = h * w;
    const float *quality = quality_all + b * (size_t)P;
    const uint8_t *mask = mask_all + b * (size_t)P;
    int32_t *parent = parent_all + b * (size_t)P;
    uint32_t *fq, *fi;
    uint8_t *st;
    if (LS) { fq = (uint32_t *)lds_raw; fi = fq + cap; st = (uint8_t *)(fi + cap); }
    else { fq = gfq + b * (size_t)cap; fi = gfi + b * (size_t)cap; st = gst + b * (size_t)P; }

    // st: 0 untouched, 1 in frontier, 2 visited.  Seed = first arg-max of q over the mask.
    unsigned long long best = 0;
    for (int p = lane; p < P; p += 64) {
        st[p] = 0;
        parent[p] = -1;
        if (mask[p]) {
            unsigned long long k = ((unsigned long long)f2key(quality[p]) << 32) | (uint32_t)(0xffffffffu - (uint32_t)p);
            if (k > best) best = k;
        }
    }
    best = wave_max_u64(best);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    if (best == 0) return;   // empty mask: everything stays NaN (shape_ftp.py:1047-1048)
    int F = 0;
    bool overflow = false;
    int cur = (int)(0xffffffffu - (uint32_t)(best & 0xffffffffu));
    bool is_seed = true;

    for (;;) {
        // ---- visit `cur`: parent = lexicographically smallest visited neighbour; push new neighbours
        int y = cur / w, x = cur - y * w;
        int np = -1;
        uint8_t s = 255;
        if (lane < 8) {
            int l = lane < 4 ? lane : lane + 1;          // skip the centre of the 3x3
            int ny = y + l / 3 - 1, nx = x + l % 3 - 1;
            if (ny >= 0 && ny < h && nx >= 0 && nx < w) { np = ny * w + nx; s = ld_st(st, np, LS); }
        }
        unsigned long long vis = __ballot(s == 2);
        int par = cur;
        if (!is_seed) {
            int pl = __ffsll((long long)vis) - 1;      // always >= 0: cur entered the frontier from a visited pixel
            par = __shfl(np, pl, 64);
        }
        bool fresh = (s == 0) && mask[np < 0 ? 0 : np] && np >= 0;
        uint32_t nk = fresh ? f2key(quality[np]) : 0u;
        unsigned long long nb = __ballot(fresh);
        int cnt = (int)__popcll(nb);
        if (F + cnt > cap) { overflow = true; break; }
        if (fresh) {
            int pos = F + (int)__popcll(nb & ((1ull << lane) - 1ull));
            fq[pos] = nk;
            fi[pos] = (uint32_t)np;
            st[np] = 1;
        }
        if (lane == 0) { parent[cur] = par; st[cur] = 2; }
        F += cnt;
        is_seed = false;
        if (F == 0) break;
        if (!LS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

        // ---- pop the frontier maximum: (q desc, index asc)
        uint32_t bq = 0, bi = 0xffffffffu;
        int bslot = -1;
        for (int j = lane; j < F; j += 64) {
            uint32_t kq = fq[j], ki = fi[j];
            if (kq > bq || (kq == bq && ki < bi)) { bq = kq; bi = ki; bslot = j; }
        }
        uint32_t mq = wave_max_u32(bq);
        unsigned long long tie = __ballot(bq == mq && bslot >= 0);
        int leader;
        if (__popcll(tie) == 1) leader = __ffsll((long long)tie) - 1;
        else {
            uint32_t mi = ~wave_max_u32((bq == mq && bslot >= 0) ? ~bi : 0u);
            unsigned long long who = __ballot(bq == mq && bslot >= 0 && bi == mi);
            leader = __ffsll((long long)who) - 1;
        }
        int slot = __shfl(bslot, leader, 64);
        cur = (int)__shfl(bi, leader, 64);
        int last = F - 1;
        if (lane == 0 && slot != last) { fq[slot] = fq[last]; fi[slot] = fi[last]; }
        F = last;
        if (!LS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    if (overflow && lane == 0) status[b] = 2;
}

// Tree -> wrap counts -> unwrapped phase, one 1024-thread workgroup per frame.
// Each pixel carries one 64-bit word (parent index | wrap-count increment relative to that parent), where
// inc = the integer n with (w[p]-w[par]) + 2*pi*n in (-pi, pi].  Pointer jumping replaces
// (par, inc) by (par[par], inc + inc[par]) IN PLACE and asynchronously: every word always states a true
// relation "k[p] = k[par] + inc" (the k are fixed by the tree), and a parent's word is read with one 64-bit
// load, so any interleaving is consistent; rounds repeat until every pixel points at the seed.
// ppar_all (optional): parents in PADDED (h+2)x(w+2) index space as written by the ranked flood kernels;
// they are converted here and stored to parent_all (the plane the parity tests read back).
__device__ inline unsigned long long ld_u64c(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_u64c(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(1024) void k_unwrap_tree(const float *__restrict__ wrapped_all, int32_t *parent_all,
                                                      const int32_t *__restrict__ ppar_in, size_t gstride, unsigned long long *words_all,
                                                      float *__restrict__ unwrapped_all, int h, int w, const int32_t *__restrict__ plain_parents, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    __shared__ int s_changed;
    const size_t b = blockIdx.x;
    // plain_parents[b] != 0: this frame's parents were written to parent_all directly (generic flood), not to the padded plane
    const int32_t *ppar_all = (plain_parents && plain_parents[b]) ? nullptr : ppar_in;
    const int P = h * w;
    const float *wrapped = wrapped_all + b * (size_t)P;
    int32_t *tree = parent_all + b * (size_t)P;
    unsigned long long *word = words_all + b * gstride;
    const double twopi = 6.283185307179586476925286766559, pi_d = 3.14159265358979323846;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        int par;
        if (ppar_all) {
            const int W2 = w + 2;
            int y = p / w, x = p - y * w;
            int pp = ppar_all[b * gstride + (size_t)(y + 1) * W2 + x + 1];
            par = pp < 0 ? -1 : (pp / W2 - 1) * w + (pp % W2 - 1);
            tree[p] = par;
        } else par = tree[p];
        int v = 0;
        if (par >= 0 && par != p) {
            double d = (double)wrapped[p] - (double)wrapped[par];
            double k = -rint(d / twopi);
            double dd = d + twopi * k;
            if (dd <= -pi_d) k += 1.0;
            else if (dd > pi_d) k -= 1.0;
            v = (int)k;
        }
        word[p] = (unsigned long long)(uint32_t)par | ((unsigned long long)(uint32_t)v << 32);
    }
    __threadfence();
    __syncthreads();
    for (int round = 0; round < 64; round++) {
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        int changed = 0;
        for (int p0 = threadIdx.x; p0 < P; p0 += 4 * blockDim.x) {
            unsigned long long wv[4], wq[4];
            int pp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { pp[u] = p0 + u * blockDim.x; wv[u] = pp[u] < P ? ld_u64c(word + pp[u]) : ~0ull; }
#pragma unroll
            for (int u = 0; u < 4; u++) { int par = (int)(uint32_t)wv[u]; wq[u] = (par >= 0 && par != pp[u]) ? ld_u64c(word + par) : ~0ull; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int par = (int)(uint32_t)wv[u];
                if (par < 0 || par == pp[u]) continue;
                int pq = (int)(uint32_t)wq[u];
                if (pq != par && pq >= 0) {
                    int v = (int)(uint32_t)(wv[u] >> 32) + (int)(uint32_t)(wq[u] >> 32);
                    st_u64c(word + pp[u], (unsigned long long)(uint32_t)pq | ((unsigned long long)(uint32_t)v << 32));
                    changed = 1;
                }
            }
        }
        if (changed) s_changed = 1;
        __threadfence();
        __syncthreads();
        int any = s_changed;
        __syncthreads();
        if (!any) break;
    }
    float *unwrapped = unwrapped_all + b * (size_t)P;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        unsigned long long wv = ld_u64c(word + p);
        float u = __uint_as_float(0x7fc00000u);
        if ((int)(uint32_t)wv >= 0) u = (float)((double)wrapped[p] + twopi * (double)(int)(uint32_t)(wv >> 32));
        unwrapped[p] = u;
    }
}

size_t unwrap_fast_scratch_bytes_per_frame(int h, int w);
static size_t unwrap_flood_scratch_bytes_per_frame(int h, int w)
{
    size_t P = (size_t)h * w, EN = (size_t)(h + 2) * (w + 2);
    const size_t code_bytes = EN > 65533 ? 4 * EN + 32 : 2 * EN + 16;     // uint32 rank codes for frames beyond the uint16 range
    return P /*st*/ + 5 * EN * sizeof(uint32_t) /*sort / frontier / jump buffers / padded parents*/ + code_bytes /*rank codes*/ + 64 /*seed, n, flag*/ + 1536 /*alignment of the sub-planes, also for a batch of one*/;
}

// the flood's planes for B frames come first, the consistency check's (k_unwrap_fast.hip) behind them
size_t unwrap_scratch_bytes_per_frame(int h, int w) { return unwrap_flood_scratch_bytes_per_frame(h, w) + unwrap_fast_scratch_bytes_per_frame(h, w) + 16; }

bool unwrap_ranked_supported(int h, int w);
bool launch_unwrap_ranked(const float *quality, const uint8_t *mask, uint32_t *g0, uint32_t *g1, uint32_t *g2, uint32_t *g3,
                          int32_t *ppar, size_t gstride, uint16_t *rank16, int32_t *seed, int32_t *status, int B, int h, int w,
                          hipStream_t st, hipEvent_t ev_flood, int flood_tier, const int32_t *need);
void launch_unwrap_replay(const float *wrapped, const uint32_t *order, size_t ostride, const int32_t *ppar, size_t gstride, int32_t *tree,
                          float *unwrapped, int B, int h, int w, hipStream_t st, const int32_t *need);
bool unwrap_big_supported(int h, int w);
bool unwrap_fast_supported(int h, int w);
void launch_unwrap_fast(const float *wrapped, const float *quality, const uint8_t *mask, float *unwrapped, int32_t *need, void *scratch, int B, int h, int w,
                        hipStream_t st);
void launch_unwrap_rank32(const float *quality, const uint8_t *mask, uint32_t *gA, uint32_t *gB, size_t gstride, uint32_t *rank32, int32_t *seed,
                          int32_t *n_out, int B, int h, int w, hipStream_t st, const int32_t *need);
void launch_unwrap_flood_big(uint32_t *code, const int32_t *seed, const int32_t *n, const uint32_t *inv, size_t inv_stride, int32_t *ppar,
                             size_t gstride, int32_t *need_generic, bool force_generic, int B, int h, int w, hipStream_t st, const int32_t *need);

static int unwrap_lds_cap(int P)
{
    long avail = 160 * 1024 - (long)((P + 15) & ~15);
    if (avail < 64 * 1024) return 0;
    return (int)((avail / 8) & ~63);
}

void launch_unwrap(const float *wrapped, const float *quality, const uint8_t *mask, float *unwrapped, int32_t *parent,
                   void *scratch, int32_t *status, int B, int h, int w, hipStream_t st, hipEvent_t ev_mid, hipEvent_t ev_flood, int flood_tier, int32_t *need_buf)
{
    int P = h * w;
    size_t n = (size_t)B * P;
    size_t EN = (size_t)(h + 2) * (w + 2);           // per-frame stride of the uint32 planes
    // scratch: [B*P st bytes][5 x B*EN u32][B*EN u16 rank codes][B seeds]
    uint8_t *gst = (uint8_t *)scratch;
    uint32_t *g0 = (uint32_t *)((uint8_t *)scratch + ((n + 255) & ~(size_t)255));
    size_t gn = (size_t)B * EN;
    uint32_t *g1 = g0 + gn, *g2 = g1 + gn, *g3 = g2 + gn, *g4 = g3 + gn;
    int cap = unwrap_lds_cap(P);
    const int32_t *ppar = nullptr;
    // First the consistency check (k_unwrap_fast.hip): frames whose wrapped field is path-independent on the seed's component get their
    // plane from a parallel integration, and every kernel of the priority flood below skips them (need[b] = 0).  The parent plane of
    // such a frame is not produced (it is a by-product of the flood; the parity tests that compare trees switch the check off).
    const int32_t *need = nullptr;
    if (need_buf && unwrap_fast_supported(h, w)) {
        void *fs = (void *)(((uintptr_t)scratch + unwrap_flood_scratch_bytes_per_frame(h, w) * (size_t)B + 255) & ~(uintptr_t)255);
        launch_unwrap_fast(wrapped, quality, mask, unwrapped, need_buf, fs, B, h, w, st);
        need = need_buf;
    }
    if (unwrap_ranked_supported(h, w)) {
        uint8_t *after = (uint8_t *)(g4 + gn);
        after = (uint8_t *)(((uintptr_t)after + 255) & ~(uintptr_t)255);
        uint16_t *rank16 = (uint16_t *)after;
        int32_t *seed = (int32_t *)(after + (((gn + 8 * (size_t)B) * 2 + 255) & ~(size_t)255));
        bool logged = launch_unwrap_ranked(quality, mask, g0, g1, g2, g3, (int32_t *)g4, EN, rank16, seed, status, B, h, w, st, ev_flood, flood_tier, need);
        ppar = (const int32_t *)g4;
        if (logged) {
            if (ev_mid) hipEventRecord(ev_mid, st);
            launch_unwrap_replay(wrapped, g2, 2 * EN, ppar, EN, parent, unwrapped, B, h, w, st, need);
            return;
        }
    } else if (unwrap_big_supported(h, w) && flood_tier >= 2) {
        // frames beyond the uint16 rank range (native crops): 32-bit ranks, bitmap priority queue in LDS, plane in global memory
        // (k_unwrap_big.hip); masks too large for the bitmap go through the generic kernel below, frame by frame
        uint8_t *after = (uint8_t *)(g4 + gn);
        after = (uint8_t *)(((uintptr_t)after + 255) & ~(uintptr_t)255);
        uint32_t *rank32 = (uint32_t *)after;
        int32_t *seed = (int32_t *)(after + (((gn + 8 * (size_t)B) * 4 + 255) & ~(size_t)255));
        int32_t *nmask = seed + B, *need_generic = nmask + B;
        launch_unwrap_rank32(quality, mask, g0, g2, EN, rank32, seed, nmask, B, h, w, st, need);
        if (ev_flood) hipEventRecord(ev_flood, st);
        launch_unwrap_flood_big(rank32, seed, nmask, g0, 2 * EN, (int32_t *)g4, EN, need_generic, flood_tier == 3, B, h, w, st, need);
        // (the generic kernel's frontier arrays reuse g0 | g1: the sorted indices are dead by now)
        hipLaunchKernelGGL(k_unwrap_flood<false>, dim3(B), dim3(64), 0, st, quality, mask, parent, gst, g0, g1, P, status, h, w, need_generic, need);
        if (ev_mid) hipEventRecord(ev_mid, st);
        hipLaunchKernelGGL(k_unwrap_tree, dim3(B), dim3(1024), 0, st, wrapped, parent, (const int32_t *)g4, EN, (unsigned long long *)g0, unwrapped, h, w,
                           need_generic, need);
        return;
    } else if (cap > 0) {
        if (ev_flood) hipEventRecord(ev_flood, st);
        static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_unwrap_flood<true>, 160 * 1024);
        size_t lds = (size_t)cap * 8 + ((P + 15) & ~15);
        hipLaunchKernelGGL(k_unwrap_flood<true>, dim3(B), dim3(64), lds, st, quality, mask, parent, gst, g0, g1, cap, status, h, w, (const int32_t *)nullptr, need);
    } else {
        if (ev_flood) hipEventRecord(ev_flood, st);
        hipLaunchKernelGGL(k_unwrap_flood<false>, dim3(B), dim3(64), 0, st, quality, mask, parent, gst, g0, g1, P, status, h, w, (const int32_t *)nullptr, need);
    }
    if (ev_mid) hipEventRecord(ev_mid, st);
    // g0|g1 (2 x B*EN uint32, contiguous) hold the per-pixel 64-bit words; stride EN words per frame
    hipLaunchKernelGGL(k_unwrap_tree, dim3(B), dim3(1024), 0, st, wrapped, parent, ppar, EN, (unsigned long long *)g0, unwrapped, h, w, (const int32_t *)nullptr, need);
}

}  // namespace vf
