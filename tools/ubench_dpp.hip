// micro-benchmark: cost of the ordered lane sums of the Telea estimator (chains of v_add_f32_dpp) with ONE wave per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CTRL, int NCH, int N>
__device__ inline float chains(float x, float (&S)[NCH])
{
#pragma unroll
    for (int t = 1; t < N; t++)
#pragma unroll
        for (int c = 0; c < NCH; c++) S[c] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(S[c]), CTRL, 0xf, 0xf, true)) + x;
    float r = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) r += S[c];
    return r;
}
template <int CTRL, int NCH>
__global__ __launch_bounds__(64) void k(unsigned long long *out, float *sink, int iters)
{
    float S[NCH];
    for (int c = 0; c < NCH; c++) S[c] = threadIdx.x * 0.5f + c;
    float x = 1.0f + threadIdx.x * 1e-3f, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { acc += chains<CTRL, NCH, 29>(x, S); asm volatile("" : "+v"(x)); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int NCH>
__global__ __launch_bounds__(64) void kplain(unsigned long long *out, float *sink, int iters)
{
    float S[NCH];
    for (int c = 0; c < NCH; c++) S[c] = threadIdx.x * 0.5f + c;
    float x = 1.0f + threadIdx.x * 1e-3f, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int t = 1; t < 29; t++)
#pragma unroll
            for (int c = 0; c < NCH; c++) { S[c] = S[c] + x; asm volatile("" : "+v"(S[c])); }
        for (int c = 0; c < NCH; c++) acc += S[c];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main()
{
    unsigned long long *d; float *s; hipMalloc(&d, 256 * 8); hipMalloc(&s, 256 * 64 * 4);
    const int iters = 2000;
    auto report = [&](const char *name, int nch) {
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256); hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : h) m += (double)v / 256;
        printf("%-28s %6.0f cycles per 28-step pass, %.2f cycles per add\n", name, m / iters, m / iters / (28.0 * nch));
    };
    hipLaunchKernelGGL((k<0x138, 4>), dim3(256), dim3(64), 0, 0, d, s, iters); report("wave_shr:1 x 4 chains", 4);
    hipLaunchKernelGGL((k<0x138, 1>), dim3(256), dim3(64), 0, 0, d, s, iters); report("wave_shr:1 x 1 chain", 1);
    hipLaunchKernelGGL((k<0x138, 8>), dim3(256), dim3(64), 0, 0, d, s, iters); report("wave_shr:1 x 8 chains", 8);
    hipLaunchKernelGGL((k<0x111, 4>), dim3(256), dim3(64), 0, 0, d, s, iters); report("row_shr:1 x 4 chains", 4);
    hipLaunchKernelGGL((k<0x111, 1>), dim3(256), dim3(64), 0, 0, d, s, iters); report("row_shr:1 x 1 chain", 1);
    hipLaunchKernelGGL((kplain<4>), dim3(256), dim3(64), 0, 0, d, s, iters); report("plain v_add x 4 chains", 4);
    hipLaunchKernelGGL((kplain<1>), dim3(256), dim3(64), 0, 0, d, s, iters); report("plain v_add x 1 chain", 1);
    return 0;
}
