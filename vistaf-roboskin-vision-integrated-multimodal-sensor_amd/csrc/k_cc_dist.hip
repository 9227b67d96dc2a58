// Mask topology kernels: thresholding, 8-connected components (union-find), largest component,
// exact 3x3 chamfer distance transform.
//   threshold_mask   roi & (q >= thr) & isfinite(q)                       (shape_ftp.py:753)
//   cc_label         cv2.connectedComponentsWithStats(connectivity=8)     (shape_ftp.py:712, :1244)
//   cc_largest       labels == 1 + argmax(areas)                           (shape_ftp.py:716-718)
//   chamfer          cv2.distanceTransform(DIST_L2, 3)                     (shape_ftp.py:725, :1309, :1312)
//
// One 1024-thread workgroup owns one frame, so every union-find word is only touched by one CU; loads
// of label words go through agent-scope relaxed atomics (served by L2, never a stale L1 line).
#include <cstdlib>
#include "kernels.hpp"

namespace vf {

__global__ void k_threshold_mask(const float *__restrict__ q, const uint8_t *__restrict__ roi, const float *__restrict__ thr,
                                 uint8_t *__restrict__ out, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    float v = q[b * (size_t)P + p];
    out[b * (size_t)P + p] = (uint8_t)(roi[p] && finitef(v) && v >= thr[b]);
}
void launch_threshold_mask(const float *q, const uint8_t *roi, const float *thr, uint8_t *out, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_threshold_mask, dim3((P + 255) / 256, B), dim3(256), 0, st, q, roi, thr, out, P);
}

// ---- union-find ---------------------------------------------------------------------------------
__device__ inline int ld_label(const int32_t *L, int i)
{
    return __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline int cc_find(const int32_t *L, int i)
{
    int p = ld_label(L, i);
    while (p != i) { i = p; p = ld_label(L, i); }
    return i;
}
__device__ inline void cc_unite(int32_t *L, int a, int b)
{
    for (;;) {
        a = cc_find(L, a);
        b = cc_find(L, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        // a < b: hang root b under a if b is still a root
        int old = atomicMin(&L[b], a);
        if (old == b) return;
        b = old;
    }
}

// Connected components (8-neighbourhood) by atomic union-find in global memory for frames too large for the LDS forest below: init, merge and
// flatten as separate launches over all pixels of the batch (one 1024-thread workgroup per frame took 45 ms per call on native 1182 x 1182 crops).  The union is an atomic "hang the larger root under the smaller" at agent scope,
// so the result -- every pixel labelled with the smallest pixel index of its component -- does not depend on which workgroup unites what when.
// init: every mask pixel points at the start of its horizontal run INSIDE its wave's 64-pixel segment (one ballot: a chain of "unite with the left
// neighbour" along a 1000-pixel run is a 1000-hop walk for every later find); merge then links a run to its left neighbour only where it
// crosses a segment boundary, and to the row above
__global__ void k_cc_init(const uint8_t *__restrict__ mask, int32_t *__restrict__ labels, int P, int w)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const bool in = p < P && mask[b * (size_t)P + (p < P ? p : 0)] != 0;
    const int x = p % w;
    const unsigned long long bits = __ballot(in);
    const unsigned long long rowstart = __ballot(x == 0);
    // lane j starts a run: masked, and lane j - 1 is not masked, or j is the first pixel of a row, or j == 0
    const unsigned long long starts = bits & (~(bits << 1) | rowstart | 1ull);
    if (p >= P) return;
    int lab = -1;
    if (in) {
        const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        lab = p - lane + (63 - __clzll((long long)upto));
    }
    labels[b * (size_t)P + p] = lab;
}
__global__ void k_cc_merge(const uint8_t *__restrict__ mask, int32_t *__restrict__ labels, int h, int w)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    int P = h * w;
    if (p >= P) return;
    const uint8_t *m = mask + b * (size_t)P;
    int32_t *L = labels + b * (size_t)P;
    if (!m[p]) return;
    int y = p / w, x = p - y * w;
    const bool left = x > 0 && m[p - 1];
    if ((threadIdx.x & 63) == 0 && left) cc_unite(L, p, p - 1);       // the run continues in the previous segment
    if (y > 0) {
        // only the contacts between runs of adjacent rows (as in k_cc_label_lds): a pixel whose left neighbour is set leaves the run above-left /
        // above to that neighbour, and the pixel above-right only starts a new contact when the pixel above is clear -- a couple of unions per
        // run instead of three per pixel (the big blob of a reliable mask made every one of them a walk to the same root)
        const bool ul = x > 0 && m[p - w - 1], up = m[p - w] != 0, ur = x < w - 1 && m[p - w + 1];
        if ((ul || up) && !left) cc_unite(L, p, ul ? p - w - 1 : p - w);
        if (ur && !up) cc_unite(L, p, p - w + 1);
    }
}
__global__ void k_cc_flatten(const uint8_t *__restrict__ mask, int32_t *__restrict__ labels, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    int32_t *L = labels + b * (size_t)P;
    if (mask[b * (size_t)P + p]) { int r = cc_find(L, p); if (r != p) atomicMin(&L[p], r); }
}

// ---- LDS-resident variant for frames of at most 65535 pixels: the whole label forest lives in LDS as one
// uint16 per pixel, so finds and unions are LDS round trips instead of L2 round trips.  LDS has no 16-bit
// atomics: the "hang root b under a" step is a 32-bit CAS on the word holding the label.
__device__ inline uint32_t cc16_min(uint16_t *L, int i, uint32_t val)
{
    uint32_t *wp = (uint32_t *)L + (i >> 1);
    const int sh = (i & 1) * 16;
    uint32_t old = *(volatile uint32_t *)wp;
    for (;;) {
        uint32_t cur = (old >> sh) & 0xffffu;
        if (cur <= val) return cur;
        uint32_t nw = (old & ~(0xffffu << sh)) | (val << sh);
        uint32_t prev = atomicCAS(wp, old, nw);
        if (prev == old) return cur;
        old = prev;
    }
}
__device__ inline int cc16_find(uint16_t *L, int i)
{
    volatile uint16_t *V = L;
    for (;;) {
        int p = V[i];
        if (p == i) return i;
        int g = V[p];
        if (g == p) return p;
        V[i] = (uint16_t)g;      // path halving (benign race: g is an ancestor of i)
        i = g;
    }
}
__device__ inline void cc16_unite(uint16_t *L, int a, int b)
{
    for (;;) {
        a = cc16_find(L, a);
        b = cc16_find(L, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        uint32_t old = cc16_min(L, b, (uint32_t)a);
        if ((int)old == b) return;
        b = (int)old;
    }
}

// MLDS: the mask plane is staged in LDS too (P bytes behind the forest), so every neighbour test is an LDS read
template <bool MLDS>
__global__ __launch_bounds__(1024) void k_cc_label_lds(const uint8_t *__restrict__ mask, int32_t *__restrict__ labels, int h, int w)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t L16[];
    size_t b = blockIdx.x;
    int P = h * w;
    const uint8_t *mg = mask + b * (size_t)P;
    uint8_t *ml = (uint8_t *)(L16 + ((P + 2 + 7) & ~7));
    if (MLDS) {
        for (int p = threadIdx.x * 4; p < P; p += blockDim.x * 4) {
            if (p + 3 < P && ((((uintptr_t)mg) & 3) == 0)) *(uint32_t *)(ml + p) = *(const uint32_t *)(mg + p);
            else for (int k = 0; k < 4 && p + k < P; k++) ml[p + k] = mg[p + k];
        }
        __syncthreads();
    }
    const uint8_t *m = MLDS ? (const uint8_t *)ml : mg;
    // Every pixel starts at the left end of its horizontal run (a prefix-max scan of the positions of the zero pixels of the row: 16
    // waves, one row at a time each), so a run is one tree from the start and only the contacts between runs of adjacent rows are left
    // to unite -- a few hundred unions per frame instead of four per pixel.  Roots are minimum pixel indices either way: same labels.
    {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwv = blockDim.x >> 6;
        for (int y = wid; y < h; y += nwv) {
            int carry = -1;                                     // position of the last zero pixel seen in the row
            for (int x0 = 0; x0 < w; x0 += 64) {
                const int x = x0 + lane;
                const bool on = x < w && m[y * w + x];
                int lz = (x < w && !on) ? x : (int)0x80000000;
                int t;
                t = __builtin_amdgcn_update_dpp((int)0x80000000, lz, 0x111, 0xf, 0xf, false); lz = t > lz ? t : lz;
                t = __builtin_amdgcn_update_dpp((int)0x80000000, lz, 0x112, 0xf, 0xf, false); lz = t > lz ? t : lz;
                t = __builtin_amdgcn_update_dpp((int)0x80000000, lz, 0x114, 0xf, 0xf, false); lz = t > lz ? t : lz;
                t = __builtin_amdgcn_update_dpp((int)0x80000000, lz, 0x118, 0xf, 0xf, false); lz = t > lz ? t : lz;
                t = __builtin_amdgcn_update_dpp((int)0x80000000, lz, 0x142, 0xa, 0xf, false); lz = t > lz ? t : lz;
                t = __builtin_amdgcn_update_dpp((int)0x80000000, lz, 0x143, 0xc, 0xf, false); lz = t > lz ? t : lz;
                lz = lz > carry ? lz : carry;
                carry = __builtin_amdgcn_readlane(lz, 63);
                if (x < w) L16[y * w + x] = on ? (uint16_t)(y * w + lz + 1) : (uint16_t)0xffffu;
            }
        }
    }
    if ((P & 1) && threadIdx.x == 0) L16[P] = 0xffffu;
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        if (!m[p]) continue;
        const int y = p / w, x = p - y * w;
        if (y == 0) continue;
        const bool left = x > 0 && m[p - 1];
        const bool ul = x > 0 && m[p - w - 1], up = m[p - w] != 0, ur = x < w - 1 && m[p - w + 1];
        // the run above-left / above: already united through the left neighbour when that one touches it too
        if ((ul || up) && !left) cc16_unite(L16, p, ul ? p - w - 1 : p - w);
        // a run that starts above-right
        if (ur && !up) cc16_unite(L16, p, p - w + 1);
    }
    __syncthreads();
    int32_t *out = labels + b * (size_t)P;
    for (int p = threadIdx.x; p < P; p += blockDim.x) out[p] = m[p] ? cc16_find(L16, p) : -1;
}

void launch_cc_label(const uint8_t *mask, int32_t *labels, int B, int h, int w, hipStream_t st)
{
    int P = h * w;
    const size_t forest = (size_t)((P + 2 + 7) & ~7) * 2;
    if (P <= 65535 && forest + (size_t)P + 16 <= 160 * 1024) {
        static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_cc_label_lds<true>, 160 * 1024);
        hipLaunchKernelGGL(k_cc_label_lds<true>, dim3(B), dim3(1024), forest + (size_t)P + 16, st, mask, labels, h, w);
        return;
    }
    if (P <= 65535 && forest <= 150 * 1024) {
        static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_cc_label_lds<false>, 160 * 1024);
        hipLaunchKernelGGL(k_cc_label_lds<false>, dim3(B), dim3(1024), forest, st, mask, labels, h, w);
        return;
    }
    const dim3 g((P + 255) / 256, B);
    hipLaunchKernelGGL(k_cc_init, g, dim3(256), 0, st, mask, labels, P, w);
    hipLaunchKernelGGL(k_cc_merge, g, dim3(256), 0, st, mask, labels, h, w);
    hipLaunchKernelGGL(k_cc_flatten, g, dim3(256), 0, st, mask, labels, P);
}

// areas per root (wave-aggregated atomics), then the largest root (ties: smallest root index = first
// in raster order = smallest OpenCV label), then out = (label == best) & and_static
__global__ __launch_bounds__(1024) void k_cc_largest(const int32_t *__restrict__ labels, int32_t *__restrict__ area,
                                                     const uint8_t *__restrict__ and_static, uint8_t *__restrict__ out, int P)
{
    __shared__ unsigned long long scratch[16];
    size_t b = blockIdx.x;
    const int32_t *L = labels + b * (size_t)P;
    int32_t *A = area + b * (size_t)P;
    int lane = threadIdx.x & 63;
    constexpr int U = 4;                    // independent loads in flight per thread (the loops are bound by memory round trips)
    const int T = blockDim.x;
    for (int p = threadIdx.x; p < P; p += T) A[p] = 0;
    __threadfence_block();      // one workgroup per frame: no agent-scope write-back of the L2
    __syncthreads();
    int Pr = ((P + U * 1024 - 1) / (U * 1024)) * (U * 1024);
    // run-length accumulation per wave: consecutive tiles mostly belong to the same (large) component, and an atomic per tile on one
    // address serialises the whole frame in L2
    int run_root = -1, run_cnt = 0;
    for (int p0 = threadIdx.x; p0 < Pr; p0 += U * T) {
        int root[U];
#pragma unroll
        for (int u = 0; u < U; u++) { int p = p0 + u * T; root[u] = p < P ? L[p] : -1; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            unsigned long long active = __ballot(root[u] >= 0);
            while (active) {
                int leader = __ffsll((long long)active) - 1;
                int r0 = __builtin_amdgcn_readlane(root[u], leader);
                unsigned long long same = __ballot(root[u] == r0);
                if (r0 == run_root) run_cnt += (int)__popcll(same);
                else {
                    if (run_root >= 0 && lane == 0) atomicAdd(&A[run_root], run_cnt);
                    run_root = r0; run_cnt = (int)__popcll(same);
                }
                active &= ~same;
            }
        }
    }
    if (run_root >= 0 && lane == 0) atomicAdd(&A[run_root], run_cnt);
    __threadfence_block();      // one workgroup per frame: no agent-scope write-back of the L2
    __syncthreads();
    unsigned long long best = 0;
    for (int p0 = threadIdx.x; p0 < P; p0 += U * T) {
        int lab[U], ar[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int p = p0 + u * T;
            lab[u] = p < P ? L[p] : -1;
            ar[u] = p < P ? __hip_atomic_load(&A[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            int p = p0 + u * T;
            if (lab[u] != p) continue;
            unsigned long long key = ((unsigned long long)(unsigned int)ar[u] << 32) | (unsigned int)(0x7fffffff - p);
            if (key > best) best = key;
        }
    }
    best = block_max_u64(best, scratch);
    int broot = (best >> 32) ? (0x7fffffff - (int)(best & 0xffffffffu)) : -2;
    for (int p0 = threadIdx.x; p0 < P; p0 += U * T) {
        int lab[U];
        uint8_t as[U];
#pragma unroll
        for (int u = 0; u < U; u++) { int p = p0 + u * T; lab[u] = p < P ? L[p] : -1; as[u] = (p < P && and_static) ? and_static[p] : (uint8_t)1; }
#pragma unroll
        for (int u = 0; u < U; u++) { int p = p0 + u * T; if (p < P) out[b * (size_t)P + p] = (uint8_t)(lab[u] == broot && as[u]); }
    }
}

// The same three steps as kernels over all pixels of the batch, for frames where one workgroup per frame leaves the chip idle (eight native
// crops: 1.46 ms in k_cc_largest): areas (run-length per wave, integer atomics), key of the largest root per frame (atomic max), output.
__global__ __launch_bounds__(256) void k_ccl_area(const int32_t *__restrict__ labels, int32_t *__restrict__ area, int P)
{
    constexpr int U = 4;
    const size_t b = blockIdx.y;
    const int32_t *L = labels + b * (size_t)P;
    int32_t *A = area + b * (size_t)P;
    const int lane = threadIdx.x & 63;
    // a wave takes U consecutive 64-pixel tiles: consecutive tiles mostly belong to the same (large) component
    const int p0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 * U) + lane;
    int root[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int p = p0 + u * 64; root[u] = p < P ? L[p] : -1; }
    int run_root = -1, run_cnt = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        unsigned long long active = __ballot(root[u] >= 0);
        while (active) {
            const int leader = __ffsll((long long)active) - 1;
            const int r0 = __builtin_amdgcn_readlane(root[u], leader);
            const unsigned long long same = __ballot(root[u] == r0);
            if (r0 == run_root) run_cnt += (int)__popcll(same);
            else {
                if (run_root >= 0 && lane == 0) atomicAdd(&A[run_root], run_cnt);
                run_root = r0; run_cnt = (int)__popcll(same);
            }
            active &= ~same;
        }
    }
    if (run_root >= 0 && lane == 0) atomicAdd(&A[run_root], run_cnt);
}
__global__ __launch_bounds__(256) void k_ccl_best(const int32_t *__restrict__ labels, const int32_t *__restrict__ area, unsigned long long *__restrict__ best, int P)
{
    const size_t b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = 0;
    if (p < P && labels[b * (size_t)P + p] == p)
        key = ((unsigned long long)(unsigned int)area[b * (size_t)P + p] << 32) | (unsigned int)(0x7fffffff - p);
    if (!__ballot(key != 0)) return;
    for (int o = 32; o; o >>= 1) { const unsigned long long v = __shfl_xor(key, o, 64); key = v > key ? v : key; }
    if ((threadIdx.x & 63) == 0) atomicMax(&best[b], key);
}
__global__ __launch_bounds__(256) void k_ccl_out(const int32_t *__restrict__ labels, const unsigned long long *__restrict__ best,
                                                 const uint8_t *__restrict__ and_static, uint8_t *__restrict__ out, int P)
{
    const size_t b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const unsigned long long k = best[b];
    const int broot = (k >> 32) ? (0x7fffffff - (int)(k & 0xffffffffu)) : -2;
    out[b * (size_t)P + p] = (uint8_t)(labels[b * (size_t)P + p] == broot && (and_static ? and_static[p] : (uint8_t)1));
}

// best: [B] scratch words (large frames only)
void launch_cc_largest(const int32_t *labels, int32_t *area_scratch, unsigned long long *best, const uint8_t *and_static,
                       uint8_t *out, int B, int P, hipStream_t st)
{
    if (best && P >= 262144 && B <= 192) {
        (void)hipMemsetAsync(area_scratch, 0, (size_t)B * P * sizeof(int32_t), st);
        (void)hipMemsetAsync(best, 0, (size_t)B * sizeof(unsigned long long), st);
        hipLaunchKernelGGL(k_ccl_area, dim3((P + 1023) / 1024, B), dim3(256), 0, st, labels, area_scratch, P);
        const dim3 g((P + 255) / 256, B);
        hipLaunchKernelGGL(k_ccl_best, g, dim3(256), 0, st, labels, area_scratch, best, P);
        hipLaunchKernelGGL(k_ccl_out, g, dim3(256), 0, st, labels, best, and_static, out, P);
        return;
    }
    hipLaunchKernelGGL(k_cc_largest, dim3(B), dim3(1024), 0, st, labels, area_scratch, and_static, out, P);
}

// ---- chamfer distance ---------------------------------------------------------------------------
constexpr int CH_INF = 1 << 20;
constexpr int CH_HV = 62587;    // cvRound(0.955  * 65536)
constexpr int CH_DG = 89738;    // cvRound(1.3693 * 65536)
constexpr int CH_DIST_MAX = 0x7fffffff >> 2;

// horizontal distance to the nearest "zero" pixel of each row, CH_INF when the row has none: one wave per row, 64 columns at a time.
// Left to right the position of the last zero pixel at or before x is a prefix maximum (DPP scan + carry across chunks), right to left the next
// one a suffix minimum; the distances are the same integers the two sequential sweeps count up (a thread per row was 0.6 ms per call on
// eight native crops: 1182 dependent, uncoalesced steps).
__global__ __launch_bounds__(256) void k_rowdist(const uint8_t *__restrict__ src, int invert, int32_t *__restrict__ g, int h, int w, int B)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * h) return;
    const uint8_t *s = src + (size_t)row * w;
    int32_t *o = g + (size_t)row * w;
    const int none = (int)0x80000000;
    int carry = none;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        const bool zero = x < w && (invert ? (s[x] != 0) : (s[x] == 0));
        int lz = zero ? x : none, t;
        t = __builtin_amdgcn_update_dpp(none, lz, 0x111, 0xf, 0xf, false); lz = t > lz ? t : lz;
        t = __builtin_amdgcn_update_dpp(none, lz, 0x112, 0xf, 0xf, false); lz = t > lz ? t : lz;
        t = __builtin_amdgcn_update_dpp(none, lz, 0x114, 0xf, 0xf, false); lz = t > lz ? t : lz;
        t = __builtin_amdgcn_update_dpp(none, lz, 0x118, 0xf, 0xf, false); lz = t > lz ? t : lz;
        t = __builtin_amdgcn_update_dpp(none, lz, 0x142, 0xa, 0xf, false); lz = t > lz ? t : lz;
        t = __builtin_amdgcn_update_dpp(none, lz, 0x143, 0xc, 0xf, false); lz = t > lz ? t : lz;
        lz = lz > carry ? lz : carry;
        carry = __builtin_amdgcn_readlane(lz, 63);
        if (x < w) o[x] = lz == none ? CH_INF : x - lz;
    }
    const int far = 0x7fffffff;
    carry = far;
    for (int x0 = ((w - 1) / 64) * 64; x0 >= 0; x0 -= 64) {
        const int x = x0 + lane;
        const bool zero = x < w && (invert ? (s[x] != 0) : (s[x] == 0));
        // suffix minimum over lanes l..63: mirror the lanes, prefix-minimum, mirror back
        int nz = zero ? x : far, t;
        nz = __builtin_amdgcn_ds_bpermute((63 - lane) << 2, nz);
        t = __builtin_amdgcn_update_dpp(far, nz, 0x111, 0xf, 0xf, false); nz = t < nz ? t : nz;
        t = __builtin_amdgcn_update_dpp(far, nz, 0x112, 0xf, 0xf, false); nz = t < nz ? t : nz;
        t = __builtin_amdgcn_update_dpp(far, nz, 0x114, 0xf, 0xf, false); nz = t < nz ? t : nz;
        t = __builtin_amdgcn_update_dpp(far, nz, 0x118, 0xf, 0xf, false); nz = t < nz ? t : nz;
        t = __builtin_amdgcn_update_dpp(far, nz, 0x142, 0xa, 0xf, false); nz = t < nz ? t : nz;
        t = __builtin_amdgcn_update_dpp(far, nz, 0x143, 0xc, 0xf, false); nz = t < nz ? t : nz;
        nz = __builtin_amdgcn_ds_bpermute((63 - lane) << 2, nz);
        nz = nz < carry ? nz : carry;
        carry = __builtin_amdgcn_readlane(nz, 0);
        if (x < w && nz != far) { const int d = nz - x; if (d < o[x]) o[x] = d; }
    }
}

// d(x,y) = min over rows y' of HV*|g-dy| + DG*min(g,dy)  (the two-pass 3x3 chamfer result, closed form)
__global__ void k_chamfer_cols(const int32_t *__restrict__ g, float *__restrict__ dist, int h, int w, int cap)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const int32_t *G = g + b * (size_t)h * w;
    int best = CH_DIST_MAX;
    for (int dy = 0; dy <= cap; dy++) {
        if ((long long)dy * CH_HV >= best) break;
        for (int s = 0; s < 2; s++) {
            if (s && !dy) break;
            int yy = s ? y + dy : y - dy;
            if (yy < 0 || yy >= h) continue;
            int gx = G[(size_t)yy * w + x];
            if (gx >= CH_INF) continue;
            int mn = gx < dy ? gx : dy, mx = gx < dy ? dy : gx;
            int c = CH_HV * (mx - mn) + CH_DG * mn;
            if (c < best) best = c;
        }
    }
    dist[b * (size_t)h * w + (size_t)y * w + x] = (float)best * (1.0f / 65536.0f);
}

// ---- the two-pass 3x3 chamfer itself (cv::distanceTransform DIST_L2, 3x3 mask: distanceTransform_3x3), one wave per frame.
// Row y of the forward pass is  d(x) = min(t(x), d(x-1) + HV)  with  t(x) = 0 on a zero pixel, else
// min(up-left + DG, up + HV, up-right + DG): t is data-parallel and the recurrence along the row is a min-plus
// prefix scan, so a wave that holds PPL consecutive columns per lane does a row in a few dozen instructions
// (lane-local scan, one DPP wave scan for the carries, one DPP shift for the neighbours across lanes).  The
// backward pass is the mirror image.  Integer arithmetic throughout: results equal the sequential loops bit for
// bit.  Rows are prefetched CH2_RING rows ahead (a single wave has nothing else to hide memory latency with).
constexpr int CH2_INF = 0x7fffffff >> 2;     // cv DIST_MAX: border / initial value of the temporary plane
constexpr int CH2_RING = 4;

__device__ inline int ch2_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }   // lane l <- l-1
__device__ inline int ch2_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }   // lane l <- l+1
// inclusive prefix minimum over lanes 0..l
__device__ inline int ch2_scan_min_up(int v)
{
    const int big = 0x7fffffff;
    int t;
    t = __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;
    return v;
}

template <int PPL>
__global__ __launch_bounds__(64) void k_chamfer2(const uint8_t *__restrict__ src_all, int invert, int32_t *__restrict__ tmp_all,
                                                 float *__restrict__ dist_all, int32_t *__restrict__ tmp2_all, float *__restrict__ dist2_all, int B,
                                                 int h, int w)
{
    // workgroups [B, 2B) (launch_chamfer_pair): the distance to the complementary set of the same frames, into the second pair of planes
    const int lane = threadIdx.x;
    size_t b = blockIdx.x;
    if (b >= (size_t)B) { b -= B; invert = !invert; tmp_all = tmp2_all; dist_all = dist2_all; }
    const size_t P = (size_t)h * w;
    const uint8_t *src = src_all + b * P;
    int32_t *tmp = tmp_all + b * P;
    float *dist = dist_all + b * P;
    const int x0 = lane * PPL;
    const int step = CH_HV * PPL;

    // ---- forward pass (top-left to bottom-right); the result plane goes to tmp
    {
        int up[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) up[j] = CH2_INF;
        uint8_t ring[CH2_RING][PPL];
#pragma unroll
        for (int r = 0; r < CH2_RING; r++)
#pragma unroll
            for (int j = 0; j < PPL; j++) ring[r][j] = (r < h && x0 + j < w) ? src[(size_t)r * w + x0 + j] : (uint8_t)0;
        for (int y0 = 0; y0 < h; y0 += CH2_RING) {
#pragma unroll
            for (int r = 0; r < CH2_RING; r++) {
                const int y = y0 + r;
                if (y >= h) break;
                const int upl = ch2_shr1(up[PPL - 1], CH2_INF), upr = ch2_shl1(up[0], CH2_INF);
                int t[PPL];
#pragma unroll
                for (int j = 0; j < PPL; j++) {
                    const bool in = x0 + j < w;
                    const bool zero = in && (invert ? ring[r][j] != 0 : ring[r][j] == 0);
                    const int a = (j > 0 ? up[j - 1] : upl) + CH_DG, c = (j < PPL - 1 ? up[j + 1] : upr) + CH_DG, u = up[j] + CH_HV;
                    int m = a < u ? a : u;
                    m = c < m ? c : m;
                    t[j] = zero ? 0 : m;
                }
                // next row of the ring
#pragma unroll
                for (int j = 0; j < PPL; j++) ring[r][j] = (y + CH2_RING < h && x0 + j < w) ? src[(size_t)(y + CH2_RING) * w + x0 + j] : (uint8_t)0;
                // d(x) = min(t(x), d(x-1) + HV): lane-local scan, carries across lanes by a wave scan of (last - step * lane)
#pragma unroll
                for (int j = 1; j < PPL; j++) { int v = t[j - 1] + CH_HV; t[j] = v < t[j] ? v : t[j]; }
                const int sc = ch2_scan_min_up(t[PPL - 1] - step * lane) + step * lane;      // final value of this lane's last column
                const int carry = ch2_shr1(sc, CH2_INF);                                     // d(x0 - 1)
#pragma unroll
                for (int j = 0; j < PPL; j++) {
                    int v = carry + CH_HV * (j + 1);
                    up[j] = v < t[j] ? v : t[j];
                    if (x0 + j < w) tmp[(size_t)y * w + x0 + j] = up[j];
                }
            }
        }
    }
    __threadfence_block();
    // ---- backward pass (bottom-right to top-left)
    {
        int dn[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) dn[j] = CH2_INF;
        int ring[CH2_RING][PPL];
#pragma unroll
        for (int r = 0; r < CH2_RING; r++)
#pragma unroll
            for (int j = 0; j < PPL; j++) ring[r][j] = (h - 1 - r >= 0 && x0 + j < w) ? tmp[(size_t)(h - 1 - r) * w + x0 + j] : CH2_INF;
        for (int y0 = 0; y0 < h; y0 += CH2_RING) {
#pragma unroll
            for (int r = 0; r < CH2_RING; r++) {
                const int y = h - 1 - (y0 + r);
                if (y < 0) break;
                const int dnl = ch2_shr1(dn[PPL - 1], CH2_INF), dnr = ch2_shl1(dn[0], CH2_INF);
                int t[PPL];
#pragma unroll
                for (int j = 0; j < PPL; j++) {
                    const int a = (j < PPL - 1 ? dn[j + 1] : dnr) + CH_DG, c = (j > 0 ? dn[j - 1] : dnl) + CH_DG, u = dn[j] + CH_HV;
                    int m = a < u ? a : u;
                    m = c < m ? c : m;
                    const int cur = x0 + j < w ? ring[r][j] : CH2_INF;
                    t[j] = m < cur ? m : cur;
                }
#pragma unroll
                for (int j = 0; j < PPL; j++) ring[r][j] = (y - CH2_RING >= 0 && x0 + j < w) ? tmp[(size_t)(y - CH2_RING) * w + x0 + j] : CH2_INF;
                // d(x) = min(t(x), d(x+1) + HV): the same scan on mirrored lanes
#pragma unroll
                for (int j = PPL - 2; j >= 0; j--) { int v = t[j + 1] + CH_HV; t[j] = v < t[j] ? v : t[j]; }
                // mirror: lane l' = 63 - l, so "lanes to the right" become a prefix
                const int ml = 63 - lane;
                int first = t[0] - step * ml;
                // prefix minimum over mirrored lanes = suffix minimum over lanes: scan the value permuted to mirrored order
                first = __builtin_amdgcn_ds_bpermute(ml << 2, first);
                first = ch2_scan_min_up(first);
                first = __builtin_amdgcn_ds_bpermute(ml << 2, first) + step * ml;            // final value of this lane's first column
                const int carry = ch2_shl1(first, CH2_INF);                                  // d(x0 + PPL)
#pragma unroll
                for (int j = 0; j < PPL; j++) {
                    int v = carry + CH_HV * (PPL - j);
                    dn[j] = v < t[j] ? v : t[j];
                    if (x0 + j < w) {
                        int o = dn[j] > CH2_INF ? CH2_INF : dn[j];
                        dist[(size_t)y * w + x0 + j] = (float)o * (1.0f / 65536.0f);
                    }
                }
            }
        }
    }
}

// ---- the same distances from the closed form, one 1024-thread workgroup per frame, exact up to `cap` rows (all the callers look at).
// The chamfer metric between two pixels is HV * (max - min) + DG * min of (|dx|, |dy|), and it grows with |dx| for a fixed |dy|, so
//   d(x, y) = min over rows y' of metric(g(y', x), |y - y'|),   g = horizontal distance to the nearest zero pixel of row y'.
// Phase 1 (rows are independent: 16 waves, a DPP max / min scan per 64-pixel chunk) leaves g in LDS as uint16; phase 2 is a short
// loop per pixel with the early exit of k_chamfer_cols.  The two-pass kernel above is one wave per frame and 2 * h dependent row steps.
constexpr uint16_t CHL_NONE = 0xffffu;
__global__ __launch_bounds__(1024) void k_chamfer_lds(const uint8_t *__restrict__ src_all, int invert, float *__restrict__ dist_all, int h, int w,
                                                      int cap)
{
    extern __shared__ uint16_t chl_g[];                  // [h * w]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t b = blockIdx.x;
    const int P = h * w;
    const uint8_t *src = src_all + b * (size_t)P;
    // phase 1: the zero pixels of a row as ballot words (the byte loads of the row are issued together: one memory round trip per
    // row); the distances to the nearest zero at or left / right of a pixel are then bit scans of those words
    constexpr int CHL_W = 8;                              // 64-pixel words per row handled in registers (w <= 512)
    const int nwords = (w + 63) >> 6;
    for (int y = wid; y < h; y += 16) {
        const uint8_t *row = src + (size_t)y * w;
        uint8_t px[CHL_W];
#pragma unroll
        for (int c = 0; c < CHL_W; c++) { const int x = c * 64 + lane; px[c] = (c < nwords && x < w) ? row[x] : (uint8_t)(invert ? 0 : 1); }
        unsigned long long Z[CHL_W];
#pragma unroll
        for (int c = 0; c < CHL_W; c++) {
            const int x = c * 64 + lane;
            Z[c] = __ballot(c < nwords && x < w && (invert ? px[c] != 0 : px[c] == 0));
        }
#pragma unroll
        for (int c = 0; c < CHL_W; c++) {
            if (c >= nwords) break;
            const int x = c * 64 + lane;
            // nearest zero at or left of x
            int dl = 0x7fff0000;
            {
                const unsigned long long zl = Z[c] & ((2ull << lane) - 1ull);
                if (zl) dl = lane - (63 - __clzll((long long)zl));
                else {
#pragma unroll
                    for (int k = 1; k < CHL_W; k++) {
                        if (c - k < 0) break;
                        if (Z[c - k]) { dl = lane + 64 * k - (63 - __clzll((long long)Z[c - k])); break; }
                    }
                }
            }
            int dr = 0x7fff0000;
            {
                const unsigned long long zr = Z[c] >> lane;
                if (zr) dr = __ffsll((long long)zr) - 1;
                else {
#pragma unroll
                    for (int k = 1; k < CHL_W; k++) {
                        if (c + k >= nwords) break;
                        if (Z[c + k]) { dr = 64 * k - lane + (__ffsll((long long)Z[c + k]) - 1); break; }
                    }
                }
            }
            const int d = dl < dr ? dl : dr;
            if (x < w) chl_g[y * w + x] = d > 0xfffe ? CHL_NONE : (uint16_t)d;
        }
    }
    __syncthreads();
    // phase 2
    float *dist = dist_all + b * (size_t)P;
    for (int p = tid; p < P; p += 1024) {
        const int y = p / w, x = p - y * w;
        int best = CH_DIST_MAX;
        for (int dy = 0; dy <= cap; dy++) {
            if ((long long)dy * CH_HV >= best) break;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (s && !dy) break;
                const int yy = s ? y + dy : y - dy;
                if (yy < 0 || yy >= h) continue;
                const int gx = (int)chl_g[yy * w + x];
                if (gx == (int)CHL_NONE) continue;
                const int mn = gx < dy ? gx : dy, mx = gx < dy ? dy : gx;
                const int c = CH_HV * (mx - mn) + CH_DG * mn;
                if (c < best) best = c;
            }
        }
        dist[p] = (float)best * (1.0f / 65536.0f);
    }
}

void launch_chamfer(const uint8_t *src, bool invert, int32_t *rowdist, float *dist, int B, int h, int w, int cap_px, hipStream_t st, bool force_twopass)
{
    // small caps (the erosion margins): a handful of rows per pixel on 16 waves; for wide bands the per-pixel loop costs more than
    // the one-wave two-pass kernel (measured at cap 46: 345 us against 200 us)
    int cap = (int)((cap_px + 2) / 0.955) + 2;
    if (cap > h) cap = h;
    if (!force_twopass && cap <= 16 && (size_t)h * w * 2 <= 150 * 1024 && w <= 512) {
        static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_chamfer_lds, 160 * 1024);
        hipLaunchKernelGGL(k_chamfer_lds, dim3(B), dim3(1024), (size_t)h * w * 2, st, src, invert ? 1 : 0, dist, h, w, cap);
        return;
    }
    (void)cap_px;
    if (w <= 256) { hipLaunchKernelGGL(k_chamfer2<4>, dim3(B), dim3(64), 0, st, src, invert ? 1 : 0, rowdist, dist, nullptr, nullptr, B, h, w); return; }
    if (w <= 512) { hipLaunchKernelGGL(k_chamfer2<8>, dim3(B), dim3(64), 0, st, src, invert ? 1 : 0, rowdist, dist, nullptr, nullptr, B, h, w); return; }
    // wider frames: closed form of the same two passes, exact up to cap_px (all that the callers look at)
    int rows = B * h;
    hipLaunchKernelGGL(k_rowdist, dim3((rows + 3) / 4), dim3(256), 0, st, src, invert ? 1 : 0, rowdist, h, w, B);
    hipLaunchKernelGGL(k_chamfer_cols, dim3((w + 255) / 256, h, B), dim3(256), 0, st, rowdist, dist, h, w, cap);
}

// distance to the zero pixels (-> dist_a) and to the non-zero pixels (-> dist_b) of the same masks.  The two-pass kernel is one wave per
// frame, so both transforms of a batch run side by side in ONE launch of 2B workgroups (the chip holds four times that many waves).
void launch_chamfer_pair(const uint8_t *src, int32_t *tmp_a, float *dist_a, int32_t *tmp_b, float *dist_b, int B, int h, int w, int cap_px,
                         hipStream_t st, bool force_twopass)
{
    int cap = (int)((cap_px + 2) / 0.955) + 2;
    if (cap > h) cap = h;
    // wide bands on wide frames (native crops, band 200 px): the closed form walks up to 2 * cap rows per pixel (0.3 ms per frame, all CUs),
    // the two-pass kernel a frame's rows once each way on one wave per frame and set (4 ms, all frames side by side): the latter from 16 frames on
    const bool two_pass = (force_twopass || cap > 16 || (size_t)h * w * 2 > 150 * 1024) && (w <= 512 || (w <= 1280 && cap > 64 && (B >= 16 || force_twopass)));
    if (!two_pass) {
        launch_chamfer(src, false, tmp_a, dist_a, B, h, w, cap_px, st, false);
        launch_chamfer(src, true, tmp_b, dist_b, B, h, w, cap_px, st, false);
        return;
    }
    if (w <= 256) hipLaunchKernelGGL(k_chamfer2<4>, dim3(2 * B), dim3(64), 0, st, src, 0, tmp_a, dist_a, tmp_b, dist_b, B, h, w);
    else if (w <= 512) hipLaunchKernelGGL(k_chamfer2<8>, dim3(2 * B), dim3(64), 0, st, src, 0, tmp_a, dist_a, tmp_b, dist_b, B, h, w);
    else hipLaunchKernelGGL(k_chamfer2<20>, dim3(2 * B), dim3(64), 0, st, src, 0, tmp_a, dist_a, tmp_b, dist_b, B, h, w);
}

__global__ void k_erode_by_dist(const float *__restrict__ dist, const uint8_t *__restrict__ src, float margin, uint8_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (uint8_t)(src[i] && dist[i] > margin);
}
void launch_erode_by_dist(const float *dist, const uint8_t *src, float margin, uint8_t *out, int B, int P, hipStream_t st)
{
    size_t n = (size_t)B * P;
    hipLaunchKernelGGL(k_erode_by_dist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dist, src, margin, out, n);
}

}  // namespace vf
