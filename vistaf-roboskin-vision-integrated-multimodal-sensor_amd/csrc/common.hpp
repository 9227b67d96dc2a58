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

// ---- wave-level reductions (xor butterflies through ds_bpermute/DPP as the compiler chooses) ----
template <typename T>
__device__ inline T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ inline uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, WAVE); v = t > v ? t : v; }
    return v;
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(v, o, WAVE); v = t > v ? t : v; }
    return v;
}
__device__ inline unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(v, o, WAVE); v = t < v ? t : v; }
    return v;
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
