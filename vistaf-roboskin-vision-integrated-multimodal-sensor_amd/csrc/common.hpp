// Shared device helpers for the gfx950 FTP kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace vf {

constexpr int WAVE = 64;

// cv::borderInterpolate BORDER_REFLECT_101
__host__ __device__ inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
// cv::borderInterpolate BORDER_REFLECT
__host__ __device__ inline int reflect_edge(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p - 1 : 2 * len - 1 - p;
    return p;
}

// order-preserving float <-> uint32 (ascending)
__device__ inline uint32_t f2key(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key2f(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__device__ inline bool finitef(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

// ---- wave-level reductions on the DPP network (no LDS round trips): quad xor 1, quad xor 2, row_half_mirror,
// row_mirror leave the row result in every lane of a 16-lane row; row_bcast15 / row_bcast31 then carry it
// across rows so that lane 63 holds the wave result, which is broadcast back through an SGPR.
template <int CTRL, int RM>
__device__ inline uint32_t dppmov(uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, RM, 0xf, false); }
template <int CTRL, int RM>
__device__ inline unsigned long long dppmov(unsigned long long old, unsigned long long v)
{
    uint32_t lo = dppmov<CTRL, RM>((uint32_t)old, (uint32_t)v), hi = dppmov<CTRL, RM>((uint32_t)(old >> 32), (uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ inline uint32_t bcast63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ inline unsigned long long bcast63(unsigned long long v)
{
    return ((unsigned long long)bcast63((uint32_t)(v >> 32)) << 32) | bcast63((uint32_t)v);
}
// RAW = bits of the value (uint32_t or unsigned long long); op combines two RAW words; id = bits of op's identity
template <typename RAW, typename F>
__device__ inline RAW wave_reduce_raw(RAW v, RAW id, F op)
{
    v = op(v, dppmov<0xB1, 0xf>(v, v));
    v = op(v, dppmov<0x4E, 0xf>(v, v));
    v = op(v, dppmov<0x141, 0xf>(v, v));
    v = op(v, dppmov<0x140, 0xf>(v, v));
    v = op(v, dppmov<0x142, 0xa>(id, v));
    v = op(v, dppmov<0x143, 0xc>(id, v));
    return bcast63(v);
}
__device__ inline uint32_t wave_sum(uint32_t v) { return wave_reduce_raw<uint32_t>(v, 0u, [](uint32_t a, uint32_t b) { return a + b; }); }
__device__ inline int wave_sum(int v) { return (int)wave_sum((uint32_t)v); }
__device__ inline unsigned long long wave_sum(unsigned long long v)
{
    return wave_reduce_raw<unsigned long long>(v, 0ull, [](unsigned long long a, unsigned long long b) { return a + b; });
}
__device__ inline float wave_sum(float v)
{
    return __uint_as_float(wave_reduce_raw<uint32_t>(__float_as_uint(v), 0u,
                                                     [](uint32_t a, uint32_t b) { return __float_as_uint(__uint_as_float(a) + __uint_as_float(b)); }));
}
__device__ inline double wave_sum(double v)
{
    return __longlong_as_double((long long)wave_reduce_raw<unsigned long long>(
        (unsigned long long)__double_as_longlong(v), 0ull, [](unsigned long long a, unsigned long long b) {
            return (unsigned long long)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
        }));
}
__device__ inline uint32_t wave_max_u32(uint32_t v) { return wave_reduce_raw<uint32_t>(v, 0u, [](uint32_t a, uint32_t b) { return b > a ? b : a; }); }
__device__ inline unsigned long long wave_max_u64(unsigned long long v)
{
    return wave_reduce_raw<unsigned long long>(v, 0ull, [](unsigned long long a, unsigned long long b) { return b > a ? b : a; });
}
__device__ inline unsigned long long wave_min_u64(unsigned long long v)
{
    return wave_reduce_raw<unsigned long long>(v, ~0ull, [](unsigned long long a, unsigned long long b) { return b < a ? b : a; });
}

// block-wide sum for blockDim.x <= 1024; `scratch` must hold 16 elements of T; result valid in all threads
template <typename T>
__device__ inline T block_sum(T v, T *scratch)
{
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = 0;
    for (int i = 0; i < nw; i++) r += scratch[i];
    return r;
}
__device__ inline unsigned long long block_max_u64(unsigned long long v, unsigned long long *scratch)
{
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max_u64(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    unsigned long long r = 0;
    for (int i = 0; i < nw; i++) r = scratch[i] > r ? scratch[i] : r;
    return r;
}
__device__ inline unsigned long long block_min_u64(unsigned long long v, unsigned long long *scratch)
{
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_min_u64(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    unsigned long long r = ~0ull;
    for (int i = 0; i < nw; i++) r = scratch[i] < r ? scratch[i] : r;
    return r;
}

// Dynamic-LDS limit of a kernel above the 64 KB default: the attribute is per device, and one process may hold sessions on several
// GPUs, so "already set" is tracked per device ordinal.
struct DynLdsOnce { bool done[64] = {}; };
inline void ensure_dyn_lds(DynLdsOnce &o, const void *fn, int bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < 64 && o.done[dev]) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (dev >= 0 && dev < 64) o.done[dev] = true;
}

// calibration curve (shape_ftp.py:682-700 / force_sensor.py:129-167), double arithmetic
struct Curve { int type; double a, b, c; };
__host__ __device__ inline double curve_eval(const Curve &cv, double v)
{
    switch (cv.type) {
    case 0: return cv.a * v;
    case 1: return cv.a * v + cv.b;
    case 2: return cv.a * v * v + cv.b * v + cv.c;
    case 3: return cv.a * (1.0 - exp(-cv.b * fmax(v, 0.0)));
    case 4: return cv.a * (exp(cv.b * fmax(v, 0.0)) - 1.0);
    default: return cv.a * ((1.0 - exp(-cv.b * fmax(v - cv.c, 0.0))) - (1.0 - exp(-cv.b * fmax(0.0 - cv.c, 0.0))));
    }
}

}  // namespace vf
