// unwrap_quality_guided (shape_ftp.py:1043-1080) without its priority queue -- when the frame allows it, which these kernels PROVE per frame.
//
// The reference grows a spanning tree from the best pixel, always taking the frontier pixel of highest quality, and sets
//     u[child] = u[parent] + wrap(w[child] - w[parent]),          i.e.  u = w + 2 pi k,  k[child] = k[parent] + c(parent -> child)
// with c(a -> b) the integer that brings w[b] - w[a] into (-pi, pi] (k_unwrap_tree's arithmetic).  Which tree it grows only matters where
// the wrapped field is inconsistent (a residue inside the mask, or around a hole of it).  If some k with k[seed] = 0 satisfies
//     k[b] - k[a] = c(a -> b)  and  c(b -> a) = -c(a -> b)        for EVERY pair of 8-adjacent pixels of the seed's component of the mask,
// then summing c along any path from the seed gives k: every spanning tree -- the quality-guided one included -- yields exactly this k, and
// u = w + 2 pi k is the reference's result (in this library's integer-count form, see k_unwrap_tree).  So, for all frames of the batch at once:
//   k_uf_count / k_uf_scan   horizontal runs of the mask per row, numbered in raster order; the seed (best quality, first in raster order);
//   k_uf_rows                k relative to the run's first pixel (a prefix sum of c along the row), the run table;
//   k_uf_edges               one edge per pair of touching runs of adjacent rows (8-adjacency) with the offset difference a contact pair implies;
//   k_uf_propagate           offsets spread from the seed's run over the edges until nothing changes (runs never reached stay NaN, as pixels the
//                            flood never reaches do) -- one workgroup per frame, run table and edges in LDS when they fit;
//   k_uf_absolute            absolute k per pixel;
//   k_uf_verify              EVERY vertical and diagonal pair is checked against the equation above (the horizontal ones hold by construction);
//                            a pair within 1e-5 of the branch cut, where c depends on the direction the tree crosses it, counts as a failure;
//   k_uf_plane               frames that passed get their plane and need[b] = 0; any other frame gets need[b] = 1 and goes through the exact
//                            priority flood (k_unwrap_rank + k_unwrap_flood_* + k_unwrap_replay / k_unwrap_tree), whose kernels skip need[b] = 0.
// The reliable mask (amplitude >= p25, closed, largest component, eroded) is residue-free on every frame looked at: the synthetic bench frames
// and all five stored photograph pairs of the reference (3.2 M pixel pairs each, 0 inconsistent).  Everything is parallel and streaming --
// ~4 passes over the wrapped plane, rows spread over the whole chip -- instead of one dependent pop per pixel on one wave per frame; the same
// kernels serve 224 x 224 and the native 1182 x 1182 crops.
#include <algorithm>
#include <cstdio>
#include "kernels.hpp"

namespace vf {

constexpr uint32_t UF_KNOWN = 0x80000000u;
constexpr int UF_CH = 4;              // 64-pixel chunks of a row in flight per wave
constexpr int UF_PASSES = 8;          // sweeps over the edge list between two barriers of the offset propagation
constexpr int UF_LDS_RUNS = 6144, UF_LDS_EDGES = 12288;      // propagation in LDS: 4 B per run + 6 B per edge = 96 KB
enum { UFC_NEDGE = 0, UFC_FAIL = 1, UFC_R = 2, UFC_N = 4 };

// per-frame planes of the check, carved out of the unwrap scratch (dead again before the flood kernels of a failed frame start)
struct UfPlanes {
    int8_t *kk;                 // [P16] k of the pixel (relative to its run, later absolute); -128 = not in the mask, -127 = never reached
    int32_t *rowbase;           // [hp] first run of each row; rowbase[h] = number of runs
    uint32_t *rstate;           // [rcap] UF_KNOWN | (offset & 0xFFFF)
    uint16_t *rs, *re, *ry;     // [rcap] first / last column, row
    uint16_t *ei, *ej;          // [ecap] run above, run below
    int16_t *ed;                // [ecap] offset[below] - offset[above]
    unsigned long long *seedkey;// [B]
    int32_t *ctl;               // [B][UFC_N]
    size_t P16, hp;
    int rcap, ecap;
};

// c(a -> b): the integer k that brings d = w[b] - w[a] into (-pi, pi], i.e. what k_unwrap_tree's float64  k = -rint(d / 2 pi) + one correction
// step yields.  The wrapped plane comes from atan2, so |d| <= 2 pi and k is -1, 0 or +1: decided here by comparing the float32 difference
// with pi (four of these per pixel: the float64 form made the check VALU-bound).  The float32 subtraction is off by at most 2^-23 |d| < 8e-7,
// so the decision can differ from the float64 one only for pairs within 8e-7 of the branch cut -- and every pair within 1e-5 of the cut is
// reported as a tie (the frame then takes the flood), which also covers the pairs exactly on the cut, where c depends on the direction the
// tree crosses them.  Wherever this function's result is USED it equals k_unwrap_tree's.  (|d| beyond 2 pi + slack: the float64 form.)
__device__ inline int uf_c(float wa, float wb, bool &tie)
{
    const float d = __fsub_rn(wb, wa), ad = fabsf(d);
    const float pi_f = 3.14159274f;
    if (ad < 6.4f) {
        tie = tie || fabsf(ad - pi_f) < 1e-5f;
        return ad > pi_f ? (d > 0.f ? -1 : 1) : 0;
    }
    const double twopi = 6.283185307179586476925286766559, pi_d = 3.14159265358979323846, inv2pi = 0.15915494309189533576888376337251;
    const double dd0 = (double)wb - (double)wa;
    double k = -rint(dd0 * inv2pi);
    double dd = dd0 + twopi * k;
    if (dd <= -pi_d) { k += 1.0; dd += twopi; }
    else if (dd > pi_d) { k -= 1.0; dd -= twopi; }
    tie = tie || fabs(fabs(dd) - pi_d) < 1e-5;
    return (int)k;
}

// inclusive prefix sum / prefix maximum (values >= 0) over the 64 lanes on the DPP network; the value of lane - 1 (lane 0: `first`)
__device__ inline uint32_t uf_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ inline uint32_t uf_scan_max(uint32_t v)
{
#define UF_MX(ctrl, rm, bc) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rm, 0xf, bc); v = o_ > v ? o_ : v; }
    UF_MX(0x111, 0xf, true) UF_MX(0x112, 0xf, true) UF_MX(0x114, 0xf, true) UF_MX(0x118, 0xf, true) UF_MX(0x142, 0xa, false) UF_MX(0x143, 0xc, false)
#undef UF_MX
    return v;
}
__device__ inline uint32_t uf_prev_lane(uint32_t v, uint32_t first, int lane)
{
    const uint32_t s = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);        // wave_shr:1
    return lane == 0 ? first : s;
}

// ---- runs per row, seed.  Every wave of the row kernels takes rw consecutive rows (small frames: fewer, longer-lived waves)
__global__ __launch_bounds__(1024) void k_uf_count(const float *__restrict__ quality_all, const uint8_t *__restrict__ mask_all, UfPlanes U, int h, int w, int rw)
{
    __shared__ unsigned long long s_best[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const size_t b = blockIdx.y;
    const size_t P = (size_t)h * w;
    unsigned long long best = 0;
    for (int r = 0; r < rw; r++) {
        const int y = (blockIdx.x * nwv + wave) * rw + r;
        if (y >= h) break;
        const float *q = quality_all + b * P + (size_t)y * w;
        const uint8_t *m = mask_all + b * P + (size_t)y * w;
        int cnt = 0;
        uint32_t last = 0;
        for (int c0 = 0; c0 < w; c0 += 64 * UF_CH) {
            uint8_t mm[UF_CH];
            float qq[UF_CH];
#pragma unroll
            for (int j = 0; j < UF_CH; j++) { const int x = c0 + 64 * j + lane; mm[j] = x < w ? m[x] : (uint8_t)0; qq[j] = x < w ? q[x] : 0.f; }
#pragma unroll
            for (int j = 0; j < UF_CH; j++) {
                const int x = c0 + 64 * j + lane;
                const uint32_t on = mm[j] ? 1u : 0u;
                const uint32_t ml = uf_prev_lane(on, last, lane);
                cnt += __popcll(__ballot(on && !ml));
                last = (uint32_t)__builtin_amdgcn_readlane((int)on, 63);
                const unsigned long long key = ((unsigned long long)f2key(qq[j]) << 32) | (uint32_t)(0x7fffffff - (y * w + x));
                if (on && key > best) best = key;
            }
        }
        if (lane == 0) U.rowbase[b * U.hp + y + 1] = cnt;
    }
    best = wave_max_u64(best);
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nwv; i++) best = s_best[i] > best ? s_best[i] : best;
        if (best) atomicMax(&U.seedkey[b], best);
    }
}

// ---- rowbase = exclusive prefix of the run counts; per-frame counters
__global__ __launch_bounds__(1024) void k_uf_scan(UfPlanes U, int h)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t b = blockIdx.x;
    int32_t *rb = U.rowbase + b * U.hp;
    if (tid == 0) { s_carry = 0; rb[0] = 0; }
    __syncthreads();
    for (int y0 = 1; y0 <= h; y0 += 1024) {
        const int y = y0 + tid;
        const int v = y <= h ? rb[y] : 0;
        const int incl = (int)uf_scan_add((uint32_t)v);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int base = s_carry;
        for (int i = 0; i < wave; i++) base += s_w[i];
        if (y <= h) rb[y] = base + incl;
        __syncthreads();
        if (tid == 1023) s_carry = base + incl;
        __syncthreads();
    }
    if (tid == 0) {
        const int R = s_carry;
        U.ctl[b * UFC_N + UFC_NEDGE] = 0;
        U.ctl[b * UFC_N + UFC_R] = R;
        U.ctl[b * UFC_N + UFC_FAIL] = R > U.rcap ? 1 : 0;
    }
}

// ---- k along the rows, relative to the first pixel of the run; run table
__global__ __launch_bounds__(1024) void k_uf_rows(const float *__restrict__ wrapped_all, const uint8_t *__restrict__ mask_all, UfPlanes U, int h, int w, int rw)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const size_t b = blockIdx.y;
    if (U.ctl[b * UFC_N + UFC_FAIL]) return;
    const size_t P = (size_t)h * w;
    const size_t ro = b * (size_t)U.rcap;
    const unsigned long long le_mask = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
    bool bad = false;
    for (int r = 0; r < rw; r++) {
    const int y = (blockIdx.x * nwv + wave) * rw + r;
    if (y >= h) break;
    const float *wr = wrapped_all + b * P + (size_t)y * w;
    const uint8_t *m = mask_all + b * P + (size_t)y * w;
    int8_t *kk = U.kk + b * U.P16 + (size_t)y * w;
    const int rb = U.rowbase[b * U.hp + y];
    uint32_t last_m = 0, last_w = 0;
    int carry_v = 0, carry_n = 0;
    for (int c0 = 0; c0 < w; c0 += 64 * UF_CH) {
        uint8_t mm[UF_CH], mr[UF_CH];
        float ww[UF_CH];
#pragma unroll
        for (int j = 0; j < UF_CH; j++) {
            const int x = c0 + 64 * j + lane;
            mm[j] = x < w ? m[x] : (uint8_t)0;
            mr[j] = x + 1 < w ? m[x + 1] : (uint8_t)0;
            ww[j] = x < w ? wr[x] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < UF_CH; j++) {
            const int x = c0 + 64 * j + lane;
            if (c0 + 64 * j >= w) break;
            const uint32_t on = mm[j] ? 1u : 0u;
            const uint32_t ml = uf_prev_lane(on, last_m, lane);
            const float wl = __uint_as_float(uf_prev_lane(__float_as_uint(ww[j]), last_w, lane));
            int v = 0;
            if (on && ml) v = uf_c(wl, ww[j], bad);
            const bool start = on && !ml;
            // k relative to the run's first pixel = prefix sum of c along the row minus its value at the run's first pixel (where c = 0)
            const uint32_t ps = uf_scan_add((uint32_t)v);
            const uint32_t sl = uf_scan_max(start ? (uint32_t)lane + 1u : 0u);               // 0: the run began in an earlier chunk
            const uint32_t pstart = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((sl ? sl - 1u : 0u) << 2), (int)ps);
            v = sl ? (int)(ps - pstart) : (int)ps + carry_v;
            const unsigned long long sb = __ballot(start);
            const int idx = rb + carry_n + __popcll(sb & le_mask) - 1;
            if (x < w) {
                if (on) {
                    bad = bad || v < -126 || v > 126;
                    kk[x] = (int8_t)v;
                    if (idx < U.rcap) {
                        if (start) { U.rs[ro + idx] = (uint16_t)x; U.ry[ro + idx] = (uint16_t)y; U.rstate[ro + idx] = 0; }
                        if (!mr[j]) U.re[ro + idx] = (uint16_t)x;
                    }
                } else kk[x] = (int8_t)-128;
            }
            carry_v = __builtin_amdgcn_readlane(v, 63);
            carry_n += __popcll(sb);
            last_m = (uint32_t)__builtin_amdgcn_readlane((int)on, 63);
            last_w = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ww[j]), 63);
        }
    }
    }
    if (__ballot(bad) && lane == 0) U.ctl[b * UFC_N + UFC_FAIL] = 1;
}

// ---- one edge per pair of touching runs of adjacent rows; the seed's run starts the propagation.  Streaming: the wave of row y holds rows
// y - 1 (U) and y (D) and looks at three pixel pairs per column x: the vertical one (U[x], D[x]), emitted at the first column of every
// stretch where both rows are in the mask (such a stretch lies in one run above and one run below), and the two diagonal ones
// (U[x-1], D[x]) / (U[x], D[x-1]), emitted only where no vertical pair touches them (the two runs then meet corner to corner and nowhere
// else).  Every pair of 8-adjacent runs gets exactly one edge; a run's number is its row's first run + the run starts seen so far.
__global__ __launch_bounds__(1024) void k_uf_edges(const float *__restrict__ wrapped_all, UfPlanes U, int h, int w, int rw)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const size_t b = blockIdx.y;
    if (U.ctl[b * UFC_N + UFC_FAIL]) return;
    const size_t P = (size_t)h * w;
    const size_t ro = b * (size_t)U.rcap, eo = b * (size_t)U.ecap;
    const unsigned long long le_mask = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
    const unsigned long long sk = U.seedkey[b];
    const int seed = sk ? 0x7fffffff - (int)(uint32_t)sk : -1, sy = seed >= 0 ? seed / w : -1, sx = seed >= 0 ? seed - sy * w : -1;
    int32_t *nedge = &U.ctl[b * UFC_N + UFC_NEDGE];
    bool bad = false;
    for (int r = 0; r < rw; r++) {
        const int y = (blockIdx.x * nwv + wave) * rw + r;
        if (y >= h) break;
        const bool up = y > 0;
        const float *wD = wrapped_all + b * P + (size_t)y * w, *wU = wD - w;
        const int8_t *kD = U.kk + b * U.P16 + (size_t)y * w, *kU = kD - w;
        const int rbD = U.rowbase[b * U.hp + y], rbU = up ? U.rowbase[b * U.hp + y - 1] : 0;
        uint32_t lmU = 0, lmD = 0, lwU = 0, lwD = 0, lkU = 0, lkD = 0, liU = 0, liD = 0;      // lane 63 of the previous chunk
        int cnU = 0, cnD = 0;
        for (int c0 = 0; c0 < w; c0 += 64 * UF_CH) {
            int8_t kd[UF_CH], ku[UF_CH];
            float wd[UF_CH], wu[UF_CH];
#pragma unroll
            for (int j = 0; j < UF_CH; j++) {
                const int x = c0 + 64 * j + lane;
                kd[j] = x < w ? kD[x] : (int8_t)-128;
                wd[j] = x < w ? wD[x] : 0.f;
                ku[j] = up && x < w ? kU[x] : (int8_t)-128;
                wu[j] = up && x < w ? wU[x] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < UF_CH; j++) {
                const int x = c0 + 64 * j + lane;
                if (c0 + 64 * j >= w) break;
                const uint32_t mD = kd[j] != -128 ? 1u : 0u, mU = ku[j] != -128 ? 1u : 0u;
                const uint32_t pmD = uf_prev_lane(mD, lmD, lane), pmU = uf_prev_lane(mU, lmU, lane);
                const unsigned long long sbD = __ballot(mD && !pmD), sbU = __ballot(mU && !pmU);
                const uint32_t iD = (uint32_t)(rbD + cnD + __popcll(sbD & le_mask) - 1), iU = (uint32_t)(rbU + cnU + __popcll(sbU & le_mask) - 1);
                const uint32_t piD = uf_prev_lane(iD, liD, lane), piU = uf_prev_lane(iU, liU, lane);
                const uint32_t kDu = (uint32_t)(int)kd[j], kUu = (uint32_t)(int)ku[j];
                const uint32_t pkD = uf_prev_lane(kDu, lkD, lane), pkU = uf_prev_lane(kUu, lkU, lane);
                const uint32_t wDu = __float_as_uint(wd[j]), wUu = __float_as_uint(wu[j]);
                const uint32_t pwD = uf_prev_lane(wDu, lwD, lane), pwU = uf_prev_lane(wUu, lwU, lane);
                if (mD && y == sy && x == sx && iD < (uint32_t)U.rcap) U.rstate[ro + iD] = UF_KNOWN | (uint32_t)(uint16_t)(int16_t)(-(int)kd[j]);
                // (run above, run below, k above, k below, w above, w below) of the up to three pairs of this column
                auto emit = [&](uint32_t i, uint32_t jn, int ka, int kb, float wa, float wb) {
                    const int c = uf_c(wa, wb, bad);
                    const int e = atomicAdd(nedge, 1);
                    if (e < U.ecap) { U.ei[eo + e] = (uint16_t)i; U.ej[eo + e] = (uint16_t)jn; U.ed[eo + e] = (int16_t)(ka + c - kb); }
                };
                if (mU && mD && !(pmU && pmD)) emit(iU, iD, (int)kUu, (int)kDu, wu[j], wd[j]);
                if (pmU && mD && !mU && !pmD) emit(piU, iD, (int)pkU, (int)kDu, __uint_as_float(pwU), wd[j]);
                if (mU && pmD && !pmU && !mD) emit(iU, piD, (int)kUu, (int)pkD, wu[j], __uint_as_float(pwD));
                cnD += __popcll(sbD); cnU += __popcll(sbU);
                lmD = (uint32_t)__builtin_amdgcn_readlane((int)mD, 63); lmU = (uint32_t)__builtin_amdgcn_readlane((int)mU, 63);
                liD = (uint32_t)__builtin_amdgcn_readlane((int)iD, 63); liU = (uint32_t)__builtin_amdgcn_readlane((int)iU, 63);
                lkD = (uint32_t)__builtin_amdgcn_readlane((int)kDu, 63); lkU = (uint32_t)__builtin_amdgcn_readlane((int)kUu, 63);
                lwD = (uint32_t)__builtin_amdgcn_readlane((int)wDu, 63); lwU = (uint32_t)__builtin_amdgcn_readlane((int)wUu, 63);
            }
        }
    }
    if (__ballot(bad) && lane == 0) U.ctl[b * UFC_N + UFC_FAIL] = 1;
}

// ---- offsets spread over the edges.  A run's word is written once (flag and offset together); a sweep may or may not see what another
// thread published in the same sweep -- later sweeps do -- so a few sweeps run between two barriers.  The run words and the edges of
// frames whose tables fit are copied into LDS first (the sweeps are dependent round trips).
__device__ __attribute__((always_inline)) inline void uf_spread(uint32_t *rstate, const uint16_t *ei, const uint16_t *ej, const int16_t *ed, int E, int *s_changed)
{
    const int tid = threadIdx.x;
    for (;;) {
        __syncthreads();
        if (tid == 0) *s_changed = 0;
        __syncthreads();
        bool ch = false;
        for (int pass = 0; pass < UF_PASSES; pass++) {
            for (int e = tid; e < E; e += 1024) {
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
        if (ch) *s_changed = 1;
        __syncthreads();
        if (!*s_changed) break;
    }
}
__global__ __launch_bounds__(1024) void k_uf_propagate(UfPlanes U)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char uf_lds[];
    __shared__ int s_changed;
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    int32_t *ctl = U.ctl + b * UFC_N;
    if (ctl[UFC_FAIL]) return;
    const int R = ctl[UFC_R], E = ctl[UFC_NEDGE];
    if (E > U.ecap) { if (tid == 0) ctl[UFC_FAIL] = 1; return; }
    uint32_t *rstate = U.rstate + b * (size_t)U.rcap;
    const uint16_t *ei = U.ei + b * (size_t)U.ecap, *ej = U.ej + b * (size_t)U.ecap;
    const int16_t *ed = U.ed + b * (size_t)U.ecap;
    if (R <= UF_LDS_RUNS && E <= UF_LDS_EDGES) {
        uint32_t *l_rs = (uint32_t *)uf_lds;
        uint16_t *l_ei = (uint16_t *)(l_rs + UF_LDS_RUNS), *l_ej = l_ei + UF_LDS_EDGES;
        int16_t *l_ed = (int16_t *)(l_ej + UF_LDS_EDGES);
        for (int r = tid; r < R; r += 1024) l_rs[r] = rstate[r];
        for (int e = tid; e < E; e += 1024) { l_ei[e] = ei[e]; l_ej[e] = ej[e]; l_ed[e] = ed[e]; }
        uf_spread(l_rs, l_ei, l_ej, l_ed, E, &s_changed);
        for (int r = tid; r < R; r += 1024) rstate[r] = l_rs[r];
    } else
        uf_spread(rstate, ei, ej, ed, E, &s_changed);
}

// ---- absolute k of every reached pixel (-127: in the mask but never reached)
__global__ __launch_bounds__(1024) void k_uf_absolute(UfPlanes U, int h, int w, int rw)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const size_t b = blockIdx.y;
    if (U.ctl[b * UFC_N + UFC_FAIL]) return;
    const uint32_t *rstate = U.rstate + b * (size_t)U.rcap;
    const unsigned long long le_mask = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
    bool bad = false;
    for (int r = 0; r < rw; r++) {
    const int y = (blockIdx.x * nwv + wave) * rw + r;
    if (y >= h) break;
    int8_t *kk = U.kk + b * U.P16 + (size_t)y * w;
    const int rb = U.rowbase[b * U.hp + y];
    uint32_t last_m = 0;
    int carry_n = 0;
    for (int c0 = 0; c0 < w; c0 += 64 * UF_CH) {
        int8_t kv[UF_CH];
#pragma unroll
        for (int j = 0; j < UF_CH; j++) { const int x = c0 + 64 * j + lane; kv[j] = x < w ? kk[x] : (int8_t)-128; }
#pragma unroll
        for (int j = 0; j < UF_CH; j++) {
            const int x = c0 + 64 * j + lane;
            if (c0 + 64 * j >= w) break;
            const uint32_t on = kv[j] != -128 ? 1u : 0u;
            const uint32_t ml = uf_prev_lane(on, last_m, lane);
            const unsigned long long sb = __ballot(on && !ml);
            const int idx = rb + carry_n + __popcll(sb & le_mask) - 1;
            if (on) {
                const uint32_t st = rstate[idx];
                int v = -127;
                if (st & UF_KNOWN) { v = (int)kv[j] + (int)(int16_t)(uint16_t)st; bad = bad || v < -126 || v > 126; }
                kk[x] = (int8_t)v;
            }
            carry_n += __popcll(sb);
            last_m = (uint32_t)__builtin_amdgcn_readlane((int)on, 63);
        }
    }
    }
    if (__ballot(bad) && lane == 0) U.ctl[b * UFC_N + UFC_FAIL] = 1;
}

// ---- every vertical and diagonal pair of the reached component
__global__ __launch_bounds__(1024) void k_uf_verify(const float *__restrict__ wrapped_all, UfPlanes U, int h, int w, int rw)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const size_t b = blockIdx.y;
    if (U.ctl[b * UFC_N + UFC_FAIL]) return;
    const size_t P = (size_t)h * w;
    bool bad = false;
    for (int r = 0; r < rw; r++) {
    const int y = (blockIdx.x * nwv + wave) * rw + r;
    if (y + 1 >= h) break;
    const float *w0 = wrapped_all + b * P + (size_t)y * w, *w1 = w0 + w;
    const int8_t *k0 = U.kk + b * U.P16 + (size_t)y * w, *k1 = k0 + w;
    for (int c0 = 0; c0 < w; c0 += 64 * UF_CH) {
        int8_t kp[UF_CH], kn[UF_CH][3];
        float wp[UF_CH], wn[UF_CH][3];
#pragma unroll
        for (int j = 0; j < UF_CH; j++) {
            const int x = c0 + 64 * j + lane;
            kp[j] = x < w ? k0[x] : (int8_t)-128;
            wp[j] = x < w ? w0[x] : 0.f;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const int xn = x + d - 1;
                const bool in = x < w && xn >= 0 && xn < w;
                kn[j][d] = in ? k1[xn] : (int8_t)-128;
                wn[j][d] = in ? w1[xn] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < UF_CH; j++) {
            if (kp[j] == -128) continue;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if (kn[j][d] == -128) continue;
                if (kp[j] == -127 && kn[j][d] == -127) continue;          // a component the seed's does not touch
                const int c = uf_c(wp[j], wn[j][d], bad);
                bad = bad || kp[j] == -127 || kn[j][d] == -127 || (int)kn[j][d] - (int)kp[j] != c;
            }
        }
    }
    }
    if (__ballot(bad) && lane == 0) U.ctl[b * UFC_N + UFC_FAIL] = 1;
}

// ---- the plane (k_unwrap_tree's final expression) of the frames that passed; need[b] for the flood kernels
__global__ __launch_bounds__(256) void k_uf_plane(const float *__restrict__ wrapped_all, float *__restrict__ unwrapped_all, UfPlanes U, int32_t *__restrict__ need, int P)
{
    const size_t b = blockIdx.y;
    const int fail = U.ctl[b * UFC_N + UFC_FAIL];
    if (blockIdx.x == 0 && threadIdx.x == 0) need[b] = fail ? 1 : 0;
    if (fail) return;
    const double twopi = 6.283185307179586476925286766559;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= P) return;
    const float *wr = wrapped_all + b * (size_t)P;
    float *out = unwrapped_all + b * (size_t)P;
    const int8_t *kk = U.kk + b * U.P16;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (p + u >= P) break;
        const int kp = kk[p + u];
        float v = __uint_as_float(0x7fc00000u);
        if (kp > -127) v = (float)((double)wr[p + u] + twopi * (double)kp);
        out[p + u] = v;
    }
}

static void uf_caps(int h, int w, int &rcap, int &ecap)
{
    (void)w;
    rcap = std::min(60000, std::max(4096, 8 * h));
    ecap = std::min(65000, 2 * rcap);
}
size_t unwrap_fast_scratch_bytes_per_frame(int h, int w)
{
    int rcap, ecap;
    uf_caps(h, w, rcap, ecap);
    const size_t P16 = ((size_t)h * w + 15) & ~(size_t)15, hp = ((size_t)h + 2 + 3) & ~(size_t)3;
    return P16 + hp * 4 + (size_t)rcap * 10 + (size_t)ecap * 6 + 8 + UFC_N * 4 + 64;
}
bool unwrap_fast_supported(int h, int w) { return h >= 2 && w >= 2 && h <= 32767 && w <= 65535 && (long long)h * w < 0x7fffffffLL; }

// scratch: unwrap_fast_scratch_bytes_per_frame(h, w) * B bytes, 16-byte aligned
void launch_unwrap_fast(const float *wrapped, const float *quality, const uint8_t *mask, float *unwrapped, int32_t *need, void *scratch, int B, int h, int w,
                        hipStream_t st)
{
    UfPlanes U;
    uf_caps(h, w, U.rcap, U.ecap);
    U.P16 = ((size_t)h * w + 15) & ~(size_t)15;
    U.hp = ((size_t)h + 2 + 3) & ~(size_t)3;
    uint8_t *p = (uint8_t *)scratch;
    U.kk = (int8_t *)p; p += U.P16 * B;
    U.rowbase = (int32_t *)p; p += U.hp * 4 * B;
    U.rstate = (uint32_t *)p; p += (size_t)U.rcap * 4 * B;
    U.rs = (uint16_t *)p; p += (size_t)U.rcap * 2 * B;
    U.re = (uint16_t *)p; p += (size_t)U.rcap * 2 * B;
    U.ry = (uint16_t *)p; p += (size_t)U.rcap * 2 * B;
    U.ei = (uint16_t *)p; p += (size_t)U.ecap * 2 * B;
    U.ej = (uint16_t *)p; p += (size_t)U.ecap * 2 * B;
    U.ed = (int16_t *)p; p += (size_t)U.ecap * 2 * B;
    p = (uint8_t *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    U.seedkey = (unsigned long long *)p; p += (size_t)8 * B;
    U.ctl = (int32_t *)p;
    (void)hipMemsetAsync(U.seedkey, 0, (size_t)8 * B, st);
    // rows per wave: enough waves to fill the chip several times over, no more (a wave that lives for one row is all dispatch overhead)
    // (measured at 224 x 224 x 256: 4 waves x 4 rows 0.31 ms, 8 x 2 0.33, 2 x 8 0.34, 4 x 1 0.36, 16 x 1 0.41 -- the shape is not what these passes cost)
    const int rw = (long long)B * h >= 65536 ? 4 : (long long)B * h >= 16384 ? 2 : 1, nwv = 4;
    const dim3 rows((h + nwv * rw - 1) / (nwv * rw), B), blk(64 * nwv);
    hipLaunchKernelGGL(k_uf_count, rows, blk, 0, st, quality, mask, U, h, w, rw);
    hipLaunchKernelGGL(k_uf_scan, dim3(B), dim3(1024), 0, st, U, h);
    hipLaunchKernelGGL(k_uf_rows, rows, blk, 0, st, wrapped, mask, U, h, w, rw);
    hipLaunchKernelGGL(k_uf_edges, rows, blk, 0, st, wrapped, U, h, w, rw);
    static DynLdsOnce lds_once;
    const size_t lds = (size_t)UF_LDS_RUNS * 4 + (size_t)UF_LDS_EDGES * 6;
    ensure_dyn_lds(lds_once, (const void *)k_uf_propagate, (int)lds);
    hipLaunchKernelGGL(k_uf_propagate, dim3(B), dim3(1024), lds, st, U);
    hipLaunchKernelGGL(k_uf_absolute, rows, blk, 0, st, U, h, w, rw);
    hipLaunchKernelGGL(k_uf_verify, rows, blk, 0, st, wrapped, U, h, w, rw);
    const int P = h * w;
    hipLaunchKernelGGL(k_uf_plane, dim3((P + 1023) / 1024, B), dim3(256), 0, st, wrapped, unwrapped, U, need, P);
}

}  // namespace vf
