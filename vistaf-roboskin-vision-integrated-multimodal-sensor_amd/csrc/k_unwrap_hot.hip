// k_unwrap_flood_hot: the growth loop of unwrap_quality_guided (shape_ftp.py:1043-1080) with an O(1)
// frontier per step, one pop per step (VISTAF_FLOOD=hot; the default is the batched k_unwrap_flood_batch).  Same contract as k_unwrap_flood_ranked (k_unwrap_rank.hip): one wavefront per
// frame, padded uint16 rank plane in LDS, parents written in padded index space.
//
// The frontier is split in two, with the invariant  every HOT entry > every COLD entry:
//   HOT   up to 64 entries (rank code << 16 | padded pixel index) kept SORTED, one per lane, in a VGPR;
//         the frontier maximum (`head`) and the smallest hot entry (`tailv`) are carried as wave-uniform
//         scalars; popping is one wave_shl DPP move, inserting is one ballot + one wave_shr DPP move.
//   COLD  a two-level bitmap over rank codes in LDS (codes are unique per frame).  Inserting is a pair
//         of fire-and-forget LDS atomic ORs issued by all inserting lanes at once; when HOT runs dry the
//         top <= 64 cold codes are pulled out of the bitmap already in descending order (no sort) and
//         their pixel indices are fetched from the rank kernel's sorted index array.
// A new entry goes to HOT only if it outranks the current HOT tail (then it outranks everything cold),
// otherwise to COLD; no knowledge of the cold maximum is needed.
//
// A single wave per CU is latency bound (measured on MI355X: 64 cycles per dependent ds_read, ~8.7 cycles per
// dependent VALU op, ~25 cycles per VALU->SALU hop), so the common "downhill" step is written to have
// one LDS round trip, two ballots and no cross-lane reduction.
#include "kernels.hpp"

namespace vf {

__device__ inline uint32_t hot_dpp_max8(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false); v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// lane i <- lane i+1 (lane 63 <- 0)
__device__ inline uint32_t wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }
// lane i <- lane i-1 (lane 0 <- 0)
__device__ inline uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }

constexpr int HOT_NW = 1024;   // 64-bit words of the code bitmap (codes < 65536)

__global__ __launch_bounds__(64) void k_unwrap_flood_hot(const uint16_t *__restrict__ rank_all, const int32_t *__restrict__ seed_in,
                                                         const uint32_t *__restrict__ inv_all, size_t inv_stride, int32_t *__restrict__ ppar_all,
                                                         size_t gstride, int32_t *status, int h, int w, const int32_t *__restrict__ need_frame)
{
    if (need_frame && !need_frame[blockIdx.x]) return;        // the consistency check settled this frame (k_unwrap_fast.hip)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    const int W2 = w + 2, EN = (h + 2) * W2;
    const int EN8 = (EN + 7) & ~7;
    uint16_t *kp = (uint16_t *)lds_raw;                                  // [EN8] pixel state / rank code
    unsigned long long *L0 = (unsigned long long *)(kp + EN8);           // [HOT_NW] cold bitmap over codes
    unsigned long long *L1 = L0 + HOT_NW;                                // [16]     one bit per L0 word
    uint32_t *stage = (uint32_t *)(L1 + 16);                             // [64]     refill staging
    int32_t *ppar = ppar_all + b * gstride;
    const uint16_t *rk = rank_all + b * (size_t)EN8;
    const uint32_t *inv = inv_all + b * inv_stride;                         // inv[rank] = padded pixel index
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    {
        const uint4 *src = (const uint4 *)rk;
        uint4 *dst = (uint4 *)kp;
        int nv = EN >> 3;
        for (int i = lane; i < nv; i += 64) dst[i] = src[i];
        for (int p = (nv << 3) + lane; p < EN; p += 64) kp[p] = rk[p];
    }
    for (int i = lane; i < HOT_NW + 16; i += 64) L0[i] = 0ull;
    for (int p = lane; p < EN; p += 64) ppar[p] = -1;
    __syncthreads();
    int cur = __builtin_amdgcn_readfirstlane(seed_in[b]);   // wave-uniform: keep the loop state in SGPRs
    if (cur < 0) return;                                                 // empty mask (shape_ftp.py:1047-1048)
    int doff;
    {
        int l = (lane & 7) < 4 ? (lane & 7) : (lane & 7) + 1;
        doff = (l / 3 - 1) * W2 + (l % 3 - 1);
    }
    uint32_t hot = 0;            // sorted descending over lanes 0..H-1, 0 elsewhere
    int H = 0;                   // entries in HOT
    int C = 0;                   // entries in COLD
    uint32_t head = 0;           // hot entry of lane 0 (uniform copy)
    uint32_t tailv = 0;          // hot entry of lane H-1 (uniform copy), 0 when H == 0
    if (lane == 0) { kp[cur] = 1; ppar[cur] = cur; }
    bool first = true;

    for (;;) {
        uint32_t v = 0;
        int np = cur + doff;
        if (lane < 8) v = kp[np];

        // ---- HOT ran dry: pull the top <= 64 cold codes (descending) out of the bitmap
        if (H == 0 && C > 0) {
            unsigned long long l1 = lane < 16 ? L1[lane] : 0ull;
            unsigned long long nz = __ballot(l1 != 0ull);
            int top1 = 63 - __clzll((long long)nz);
            uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)l1, top1);
            uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(l1 >> 32), top1);
            unsigned long long w1 = ((unsigned long long)hi << 32) | lo;
            int wtop = top1 * 64 + (63 - __clzll((long long)w1));
            int wi = wtop - lane;
            unsigned long long word = wi >= 0 ? L0[wi] : 0ull;
            int cnt = (int)__popcll(word);
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            int pre = incl - cnt;
            int take = 64 - pre;
            take = take < 0 ? 0 : (take > cnt ? cnt : take);
            int total = __builtin_amdgcn_readlane(incl, 63);
            total = total > 64 ? 64 : total;
            unsigned long long rem = word;
            int maxtake = take;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(maxtake, o, 64); maxtake = t > maxtake ? t : maxtake; }
            maxtake = __builtin_amdgcn_readfirstlane(maxtake);
            for (int j = 0; j < maxtake; j++) {
                if (j < take) {
                    int bit = 63 - __clzll((long long)rem);
                    stage[pre + j] = (uint32_t)(wi * 64 + bit);
                    rem &= ~(1ull << bit);
                }
            }
            if (take > 0) {
                L0[wi] = rem;
                if (rem == 0ull) atomicAnd(&L1[wi >> 6], ~(1ull << (wi & 63)));
            }
            uint32_t code = lane < total ? stage[lane] : 0u;
            uint32_t idx = lane < total ? inv[code - 3u] : 0u;
            hot = lane < total ? ((code << 16) | idx) : 0u;
            H = total;
            C -= total;
            head = (uint32_t)__builtin_amdgcn_readfirstlane((int)hot);
            tailv = (uint32_t)__builtin_amdgcn_readlane((int)hot, total - 1);
        }

        const bool fresh = v >= 3;
        const uint32_t e = fresh ? ((v << 16) | (uint32_t)np) : 0u;      // new frontier entries of lanes 0..7
        const unsigned long long visb = __ballot(v == 1);
        unsigned long long app = __ballot(fresh);
        const unsigned long long beat = __ballot(e > head);
        // parent = lexicographically smallest visited neighbour = lowest lane with v == 1
        if (!first && v == 1 && (visb & lt_mask) == 0ull) ppar[cur] = np;
        first = false;

        int next;
        if (beat) {                                                      // a new neighbour outranks the whole frontier
            uint32_t nm = hot_dpp_max8(e);
            next = (int)(nm & 0xffffu);
            app &= ~__ballot(fresh && e == nm);
        } else {
            if (head == 0u) break;                                       // frontier exhausted
            next = (int)(head & 0xffffu);
            hot = wave_shl1(hot);
            H--;
            head = (uint32_t)__builtin_amdgcn_readfirstlane((int)hot);
            if (H == 0) tailv = 0u;
        }

        // ---- remaining new entries: HOT when they outrank the HOT tail (or the frontier is empty), else COLD
        unsigned long long hotb = (H > 0) ? __ballot(((app >> lane) & 1ull) && e > tailv)
                                          : ((C == 0) ? app : 0ull);
        unsigned long long coldb = app & ~hotb;
        while (hotb) {
            int l = __ffsll((long long)hotb) - 1;
            hotb &= hotb - 1ull;
            uint32_t el = (uint32_t)__builtin_amdgcn_readlane((int)e, l);
            if (H > 0 && !(el > tailv)) { coldb |= 1ull << l; continue; }   // the tail moved up meanwhile
            int pos = (int)__popcll(__ballot(hot > el));
            uint32_t displaced = (uint32_t)__builtin_amdgcn_readlane((int)hot, 63);
            uint32_t sh = wave_shr1(hot);
            hot = lane < pos ? hot : (lane == pos ? el : sh);
            if (pos == 0) head = el;
            if (H == 0) tailv = el;
            if (H < 64) H++;
            else {
                // HOT was full: its old tail falls into COLD
                uint32_t tc = displaced >> 16;
                if (lane == 0) {
                    atomicOr(&L0[tc >> 6], 1ull << (tc & 63u));
                    atomicOr(&L1[tc >> 12], 1ull << ((tc >> 6) & 63u));
                }
                C++;
                tailv = (uint32_t)__builtin_amdgcn_readlane((int)hot, 63);
            }
        }
        if ((coldb >> lane) & 1ull) {
            uint32_t c = e >> 16;
            atomicOr(&L0[c >> 6], 1ull << (c & 63u));
            atomicOr(&L1[c >> 12], 1ull << ((c >> 6) & 63u));
        }
        C += (int)__popcll(coldb);
        if ((app >> lane) & 1ull) kp[np] = 2;
        cur = next;
        if (lane == 0) kp[cur] = 1;
    }
    (void)status;
}

bool unwrap_hot_supported(int h, int w)
{
    long EN = (long)(h + 2) * (w + 2);
    long lds = (((EN + 7) & ~7L)) * 2 + (HOT_NW + 16) * 8 + 256;
    return EN <= 65533 && lds <= 160 * 1024;
}

void launch_unwrap_flood_hot(const uint16_t *rank16, const int32_t *seed, const uint32_t *inv, size_t inv_stride, int32_t *ppar, size_t gstride,
                             int32_t *status, int B, int h, int w, hipStream_t st, const int32_t *need)
{
    long EN = (long)(h + 2) * (w + 2);
    size_t lds = (size_t)(((EN + 7) & ~7L)) * 2 + (HOT_NW + 16) * 8 + 256;
    static DynLdsOnce lds_once;
        ensure_dyn_lds(lds_once, (const void *)k_unwrap_flood_hot, 160 * 1024);
    hipLaunchKernelGGL(k_unwrap_flood_hot, dim3(B), dim3(64), lds, st, rank16, seed, inv, inv_stride, ppar, gstride, status, h, w, need);
}

}  // namespace vf
