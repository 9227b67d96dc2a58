// micro-benchmark: shader clock and dependent LDS-read latency with ONE wave per CU on 256 CUs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int iters)
{
    __shared__ unsigned short a[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) a[i] = (unsigned short)((i * 7 + 13) & 4095);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned v = threadIdx.x;
    for (int i = 0; i < iters; i++) v = a[v];             // dependent LDS chain
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned w = v;
    for (int i = 0; i < iters; i++) { w = w * 1664525u + 1013904223u; w ^= w >> 7; w += 3; w ^= w << 3; }   // 6 dependent VALU / iter
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    unsigned long long bal = 0;
    for (int i = 0; i < iters; i++) { bal += __popcll(__ballot((w + i) & 1)); }
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 8 + 0] = t1 - t0; out[blockIdx.x * 8 + 1] = r1 - r0; out[blockIdx.x * 8 + 2] = t2 - t1; out[blockIdx.x * 8 + 3] = t3 - t2; out[blockIdx.x * 8 + 4] = v + w + bal; }
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 256 * 8 * 8);
    int iters = 200000;
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, d, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0, lds = 0, valu = 0, bal = 0;
    for (int b = 0; b < 256; b++) { clk += (double)h[b*8] / ((double)h[b*8+1] / 100.0); lds += (double)h[b*8] / iters; valu += (double)h[b*8+2] / iters / 6.0; bal += (double)h[b*8+3] / iters; }
    printf("shader clock %.1f MHz | dependent ds_read_u16 %.1f cyc | dependent VALU op %.2f cyc | ballot+popc+add iter %.1f cyc\n", clk / 256, lds / 256, valu / 256, bal / 256);
    return 0;
}
