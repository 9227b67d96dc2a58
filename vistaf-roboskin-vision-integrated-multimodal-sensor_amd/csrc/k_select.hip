// Per-frame exact percentiles / medians (np.percentile, np.nanpercentile, np.nanmedian call sites:
// shape_ftp.py:347, :354, :622, :846, :1720-1732, :1746, :1760, :1764).  One 1024-thread workgroup
// per frame; see select.hpp for the algorithm.
#include "kernels.hpp"
#include "select.hpp"

namespace vf {

struct PlaneGetter {
    const float *v; const uint8_t *m; float le; bool use_le, use_abs;
    __device__ bool operator()(int i, uint32_t &key) const
    {
        // both loads unconditional, so that the SEL_U elements of a batch are all in flight together (select.hpp: sel_foreach)
        const uint8_t mk = m[i];
        float x = v[i];
        bool ok = mk != 0 && finitef(x);
        if (use_abs) x = fabsf(x);
        if (use_le && !(x <= le)) ok = false;
        key = f2key(x);
        return ok;
    }
};

__global__ __launch_bounds__(SEL_T) void k_select(const float *__restrict__ vals, const uint8_t *__restrict__ mask, size_t mask_stride,
                                                  const float *__restrict__ le_thr, int use_abs, const float *__restrict__ reqs,
                                                  int nreq, float *__restrict__ out, int *__restrict__ counts, int P)
{
    __shared__ SelShared sh;
    size_t b = blockIdx.x;
    PlaneGetter g{vals + b * (size_t)P, mask + b * mask_stride, le_thr ? le_thr[b] : 0.f, le_thr != nullptr, use_abs != 0};
    uint32_t n, kmin, kmax;
    block_minmax(g, P, sh, n, kmin, kmax);
    if (threadIdx.x == 0 && counts) counts[b] = (int)n;
    for (int j = 0; j < nreq; j++) {
        float q = reqs[j];
        float r = (q < 0.f) ? block_median(g, P, sh, n, kmin, kmax) : block_percentile(g, P, q, sh, n, kmin, kmax);
        if (threadIdx.x == 0) out[b * (size_t)nreq + j] = r;
        __syncthreads();
    }
}

void launch_select(const float *vals, const uint8_t *mask, size_t mask_stride, const float *le_thr, bool use_abs,
                   const float *reqs_dev, int nreq, float *out, int *counts, int B, int P, hipStream_t st, void *big_scratch)
{
    if (big_scratch && nreq <= 4 && big_frames(B, P)) {          // large frames: every sweep over all pixels of the batch (k_big.hip)
        launch_select_big(vals, mask, mask_stride, le_thr, use_abs, reqs_dev, nreq, out, counts, B, P, big_scratch, st);
        return;
    }
    hipLaunchKernelGGL(k_select, dim3(B), dim3(SEL_T), 0, st, vals, mask, mask_stride, le_thr, use_abs ? 1 : 0, reqs_dev, nreq,
                       out, counts, P);
}

}  // namespace vf
