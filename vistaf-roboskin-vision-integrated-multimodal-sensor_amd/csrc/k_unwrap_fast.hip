// unwrap_quality_guided (shape_ftp.py:1043-1080) without its priority queue -- when the frame allows it, which the kernel PROVES per frame.
//
// The reference grows a spanning tree from the best pixel, always taking the frontier pixel of highest quality, and sets
//     u[child] = u[parent] + wrap(w[child] - w[parent]),          i.e.  u = w + 2 pi k,  k[child] = k[parent] + c(parent -> child)
// with c(a -> b) the integer that brings w[b] - w[a] into (-pi, pi] (k_unwrap_tree's arithmetic).  Which tree it grows only matters where
// the wrapped field is inconsistent (a residue inside the mask, or around a hole of it).  If some k with k[seed] = 0 satisfies
//     k[b] - k[a] = c(a -> b)  and  c(b -> a) = -c(a -> b)        for EVERY pair of 8-adjacent pixels of the seed's component of the mask,
// then summing c along any path from the seed gives k: every spanning tree -- the quality-guided one included -- yields exactly this k, and
// u = w + 2 pi k is the reference's result (in this library's integer-count form, see k_unwrap_tree).  So:
//   1. rows: horizontal runs of the mask, k relative to the run's first pixel (a segmented scan of c along the row);
//   2. one edge per pair of touching runs of adjacent rows (8-adjacency), with the offset difference a contact pixel pair implies;
//   3. offsets spread from the seed's run over the edges until nothing changes (runs never reached stay NaN, as pixels the flood never
//      reaches do);
//   4. EVERY vertical and diagonal pixel pair is checked against the equation above (the horizontal ones hold by construction); a tie
//      (w[b] - w[a] + 2 pi k = +-pi exactly, where c depends on the direction the tree crosses the pair) counts as a failure.
// A frame that passes gets its plane written here and need[b] = 0; any other frame gets need[b] = 1 and goes through the exact
// priority flood (k_unwrap_rank + k_unwrap_flood_* + k_unwrap_replay / k_unwrap_tree), whose kernels skip the frames with need[b] = 0.
// The reliable mask (amplitude >= p25, closed, largest component, eroded) is residue-free on every frame looked at: the synthetic bench
// frames and all five stored photograph pairs of the reference (3.2 M pixel pairs each, 0 inconsistent).
// Everything is parallel and streaming: ~3 passes over the frame's wrapped plane instead of 28 k dependent pops.
#include "kernels.hpp"

namespace vf {

constexpr int UF_T = 1024, UF_W = UF_T / 64;
constexpr int UF_PASSES = 8;          // sweeps over the edge list between two barriers of the offset propagation
constexpr uint32_t UF_KNOWN = 0x80000000u;

// c(a -> b) as k_unwrap_tree forms it; tie: the pair sits exactly on the branch cut
__device__ inline int uf_c(float wa, float wb, bool &tie)
{
    const double twopi = 6.283185307179586476925286766559, pi_d = 3.14159265358979323846;
    const double d = (double)wb - (double)wa;
    double k = -rint(d / twopi);
    const double dd = d + twopi * k;
    tie = tie || dd == pi_d || dd == -pi_d;
    if (dd <= -pi_d) k += 1.0;
    else if (dd > pi_d) k -= 1.0;
    return (int)k;
}

__global__ __launch_bounds__(UF_T) void k_unwrap_fast(const float *__restrict__ wrapped_all, const float *__restrict__ quality_all,
                                                      const uint8_t *__restrict__ mask_all, float *__restrict__ unwrapped_all, int32_t *__restrict__ need,
                                                      int h, int w, int run_cap, int edge_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char uf_lds[];
    __shared__ unsigned long long s_red[UF_W];
    __shared__ int s_fail, s_nedge, s_changed;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t b = blockIdx.x;
    const int P = h * w;
    const float *wr = wrapped_all + b * (size_t)P, *q = quality_all + b * (size_t)P;
    const uint8_t *m = mask_all + b * (size_t)P;
    float *out = unwrapped_all + b * (size_t)P;
    int8_t *kk = (int8_t *)uf_lds;                                       // [P] k of the pixel: relative to its run, later absolute; -128 = not reached
    int *rowbase = (int *)(uf_lds + (((size_t)P + 15) & ~(size_t)15));   // [h + 1] first run of each row
    uint32_t *rstate = (uint32_t *)(rowbase + ((h + 4) & ~3));           // [run_cap] UF_KNOWN | (offset & 0xFFFF)
    uint16_t *rs = (uint16_t *)(rstate + run_cap), *re = rs + run_cap, *ry = re + run_cap;      // [run_cap] first / last column, row
    uint16_t *ei = ry + run_cap, *ej = ei + edge_cap;                    // [edge_cap] run above, run below
    int16_t *ed = (int16_t *)(ej + edge_cap);                            // [edge_cap] offset[below] - offset[above]
    const unsigned long long le_mask = lane == 63 ? ~0ull : (2ull << lane) - 1ull;

    // ---- seed: highest quality in the mask, first in raster order among equals (np.argmax)
    unsigned long long best = 0;
    for (int p0 = tid; p0 < P; p0 += UF_T * 4) {
        uint8_t mm[4];
        float qq[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int p = p0 + u * UF_T; mm[u] = p < P ? m[p] : (uint8_t)0; qq[u] = p < P ? q[p] : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + u * UF_T;
            const unsigned long long key = ((unsigned long long)f2key(qq[u]) << 32) | (uint32_t)(0x7fffffff - p);
            if (mm[u] && key > best) best = key;
        }
    }
    best = wave_max_u64(best);
    if (lane == 0) s_red[wave] = best;
    if (tid == 0) { s_fail = 0; s_nedge = 0; }
    __syncthreads();
    best = 0;
    for (int i = 0; i < UF_W; i++) best = s_red[i] > best ? s_red[i] : best;
    if (best == 0) {                                       // empty mask: nothing is reached
        for (int p = tid; p < P; p += UF_T) out[p] = __uint_as_float(0x7fc00000u);
        if (tid == 0) need[b] = 0;
        return;
    }
    const int seed = 0x7fffffff - (int)(uint32_t)best, sy = seed / w, sx = seed - sy * w;

    // ---- 1a. runs per row
    for (int y = wave; y < h; y += UF_W) {
        int cnt = 0;
        for (int c0 = 0; c0 < w; c0 += 64) {
            const int x = c0 + lane;
            const bool in = x < w;
            const bool mm = in && m[y * w + x], ml = in && x > 0 && m[y * w + x - 1];
            cnt += __popcll(__ballot(mm && !ml));
        }
        if (lane == 0) rowbase[y + 1] = cnt;
    }
    if (tid == 0) rowbase[0] = 0;
    __syncthreads();
    if (wave == 0) {
        int carry = 0;
        for (int y0 = 1; y0 <= h; y0 += 64) {
            const int y = y0 + lane;
            const int v = y <= h ? rowbase[y] : 0;
            int s = v;
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(s, o, 64); if (lane >= o) s += u; }
            if (y <= h) rowbase[y] = carry + s;
            carry += __shfl(s, 63, 64);
        }
    }
    __syncthreads();
    const int R = rowbase[h];
    if (R > run_cap) { if (tid == 0) need[b] = 1; return; }
    for (int r = tid; r < R; r += UF_T) rstate[r] = 0;

    // ---- 1b. k along the rows, relative to the first pixel of the run; run table
    bool bad = false;
    for (int y = wave; y < h; y += UF_W) {
        int carry_v = 0, carry_n = 0;
        const int rb = rowbase[y];
        for (int c0 = 0; c0 < w; c0 += 64) {
            const int x = c0 + lane, p = y * w + x;
            const bool in = x < w;
            const bool mm = in && m[p], ml = in && x > 0 && m[p - 1], mr = in && x + 1 < w && m[p + 1];
            int v = 0;
            if (mm && ml) v = uf_c(wr[p - 1], wr[p], bad);
            const bool start = mm && !ml;
            // segmented inclusive scan of v, a new segment at every run start
            bool fl = start;
            for (int o = 1; o < 64; o <<= 1) {
                const int uv = __shfl_up(v, o, 64);
                const int uf = __shfl_up((int)fl, o, 64);
                if (lane >= o) { if (!fl) v += uv; fl = fl || uf; }
            }
            if (!fl) v += carry_v;                           // the run began in an earlier chunk
            const unsigned long long sb = __ballot(start);
            const int idx = rb + carry_n + __popcll(sb & le_mask) - 1;
            if (mm) {
                bad = bad || v < -127 || v > 127;
                kk[p] = (int8_t)v;
                if (start) { rs[idx] = (uint16_t)x; ry[idx] = (uint16_t)y; }
                if (!mr) re[idx] = (uint16_t)x;
            } else if (in) kk[p] = 0;
            carry_v = __shfl(v, 63, 64);
            carry_n += __popcll(sb);
        }
    }
    __syncthreads();

    // ---- 2. one edge per pair of touching runs of adjacent rows; the seed's run starts the propagation
    for (int j = tid; j < R; j += UF_T) {
        const int y = ry[j], sj = rs[j], ej_ = re[j];
        if (y == sy && sj <= sx && sx <= ej_) __hip_atomic_store(&rstate[j], UF_KNOWN | (uint32_t)(uint16_t)(int16_t)(-(int)kk[seed]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (y == 0) continue;
        for (int i = rowbase[y - 1]; i < rowbase[y]; i++) {
            const int si = rs[i], ei_ = re[i];
            if (si > ej_ + 1 || ei_ < sj - 1) continue;
            // contact pair: (y - 1, xi) in run i, (y, xj) in run j
            int xi, xj;
            const int lo = max(si, sj), hi = min(ei_, ej_);
            if (lo <= hi) xi = xj = lo;
            else if (ei_ < sj) { xi = ei_; xj = sj; }
            else { xi = si; xj = ej_; }
            const int pa = (y - 1) * w + xi, pb = y * w + xj;
            const int c = uf_c(wr[pa], wr[pb], bad);
            const int e = atomicAdd(&s_nedge, 1);
            if (e < edge_cap) { ei[e] = (uint16_t)i; ej[e] = (uint16_t)j; ed[e] = (int16_t)((int)kk[pa] + c - (int)kk[pb]); }
        }
    }
    __syncthreads();
    const int E = s_nedge;
    if (E > edge_cap) { if (tid == 0) need[b] = 1; return; }

    // ---- 3. offsets spread over the edges.  A run's word is written once (flag and offset together); a sweep may or may not see what
    // another thread published in the same sweep -- later sweeps do -- so a few sweeps run between two barriers
    for (;;) {
        __syncthreads();
        if (tid == 0) s_changed = 0;
        __syncthreads();
        bool ch = false;
        for (int pass = 0; pass < UF_PASSES; pass++) {
            for (int e = tid; e < E; e += UF_T) {
                const int i = ei[e], j = ej[e];
                const uint32_t a = __hip_atomic_load(&rstate[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t c = __hip_atomic_load(&rstate[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((a ^ c) & UF_KNOWN) {
                    const int d = ed[e];
                    if (a & UF_KNOWN) __hip_atomic_store(&rstate[j], UF_KNOWN | (uint32_t)(uint16_t)(int16_t)((int)(int16_t)(uint16_t)a + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(&rstate[i], UF_KNOWN | (uint32_t)(uint16_t)(int16_t)((int)(int16_t)(uint16_t)c - d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ch = true;
                }
            }
        }
        if (ch) s_changed = 1;
        __syncthreads();
        if (!s_changed) break;
    }

    // ---- absolute k of every reached pixel (wave per run)
    for (int r = wave; r < R; r += UF_W) {
        const uint32_t st = rstate[r];
        const int y = ry[r], s = rs[r], e = re[r], off = (int)(int16_t)(uint16_t)st;
        for (int x = s + lane; x <= e; x += 64) {
            const int p = y * w + x;
            int v = -128;
            if (st & UF_KNOWN) { v = (int)kk[p] + off; bad = bad || v < -127 || v > 127; }
            kk[p] = (int8_t)v;
        }
    }
    __syncthreads();

    // ---- 4. every vertical and diagonal pair of the reached component
    for (int p = tid; p < P; p += UF_T) {
        if (!m[p]) continue;
        const int kp = kk[p];
        const int y = p / w, x = p - y * w;
        if (y + 1 >= h) continue;
        const float wv = wr[p];
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int xn = x + dx;
            if (xn < 0 || xn >= w) continue;
            const int pn = p + w + dx;
            if (!m[pn]) continue;
            const int kn = kk[pn];
            if (kp == -128 && kn == -128) continue;               // a component the seed's does not touch
            const int c = uf_c(wv, wr[pn], bad);
            bad = bad || kp == -128 || kn == -128 || kn - kp != c;
        }
    }
    if (__ballot(bad) && lane == 0) s_fail = 1;
    __syncthreads();
    if (s_fail) { if (tid == 0) need[b] = 1; return; }

    // ---- the plane (k_unwrap_tree's final expression)
    const double twopi = 6.283185307179586476925286766559;
    for (int p = tid; p < P; p += UF_T) {
        float u = __uint_as_float(0x7fc00000u);
        if (m[p]) { const int kp = kk[p]; if (kp != -128) u = (float)((double)wr[p] + twopi * (double)kp); }
        out[p] = u;
    }
    if (tid == 0) need[b] = 0;
}

static size_t uf_lds_bytes(int h, int w, int run_cap, int edge_cap)
{
    return (((size_t)h * w + 15) & ~(size_t)15) + (size_t)((h + 4) & ~3) * 4 + (size_t)run_cap * 10 + (size_t)edge_cap * 6;
}

// caps chosen so that frames of up to 64 K pixels fit one CU's LDS; returns false when the frame is too large for this kernel
bool unwrap_fast_supported(int h, int w)
{
    return (size_t)h * w <= 65536 && h <= 1024 && w <= 65535 && uf_lds_bytes(h, w, 2048, 3072) <= 160 * 1024 - 256;
}

void launch_unwrap_fast(const float *wrapped, const float *quality, const uint8_t *mask, float *unwrapped, int32_t *need, int B, int h, int w, hipStream_t st)
{
    const int run_cap = 2048, edge_cap = 3072;
    static DynLdsOnce lds_once;
    ensure_dyn_lds(lds_once, (const void *)k_unwrap_fast, 160 * 1024 - 256);      // (+ 140 B static)
    hipLaunchKernelGGL(k_unwrap_fast, dim3(B), dim3(UF_T), uf_lds_bytes(h, w, run_cap, edge_cap), st, wrapped, quality, mask, unwrapped, need, h, w, run_cap, edge_cap);
}

}  // namespace vf
