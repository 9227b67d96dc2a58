// Internal holes of the reliable region (shape_ftp.py:1153-1204, :1770-1801).
//
// Only reachable when RELIABLE_SMOOTH_SIGMA_PX is 0: with a positive sigma masked_gaussian_smooth (:1139-1147) leaves a finite value at
// every pixel, so `known_height` equals `reliable` and compute_internal_holes_within_mask returns at its first test.  Without the smoothing
// the reliable pixels the unwrap never reached stay NaN (erode_by_distance runs AFTER largest_connected_component, :770-773, so the eroded
// mask can fall apart and the flood only covers the seed's component), and upstream
//   * marks as hole candidates the NaN pixels with >= HOLE_KNOWN_FRACTION known pixels in their k x k window (cv2.boxFilter, unnormalised,
//     BORDER_REFLECT_101) that lie >= HOLE_MIN_DIST_FROM_RELIABLE_EDGE_PX inside the reliable mask (3x3 chamfer distance);
//   * fills them by Telea inpainting (radius INPAINT_RADIUS) of a plane that holds the known heights, the median of the known heights at
//     the other NaN pixels of the reliable mask and the median of that plane everywhere else;
//   * drops the remaining NaN pixels from `output_reliable`.
// The kernels below are the glue; medians, the distance transform and the march are the path's own kernels (k_select, k_chamfer*, k_telea*).
#include "kernels.hpp"

namespace vf {

__device__ inline float hl_nan() { return __uint_as_float(0x7fc00000u); }

// height_map = phase_zeroed = detrended - bg_med (:1750-1753): NaN wherever the detrended phase is (outside reliable, unreached pixels)
__global__ void k_zeroed_keep_nan(const float *__restrict__ detr, const float *__restrict__ bg_med, const uint8_t *__restrict__ reliable,
                                  float *__restrict__ hmap, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    hmap[i] = reliable[i] ? __fsub_rn(detr[i], bg_med[b]) : hl_nan();
}
void launch_zeroed_keep_nan(const float *detr, const float *bg_med, const uint8_t *reliable, float *hmap, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_zeroed_keep_nan, dim3((P + 255) / 256, B), dim3(256), 0, st, detr, bg_med, reliable, hmap, P);
}

// compute_internal_holes_within_mask (:1153-1175): container = reliable, known = reliable & finite(height)
__global__ void k_hole_candidates(const float *__restrict__ hmap, const uint8_t *__restrict__ reliable, const float *__restrict__ dist, int ksize,
                                  float frac_thr, float min_dist, uint8_t *__restrict__ cand, int h, int w)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const size_t base = b * (size_t)h * w;
    const size_t i = base + (size_t)y * w + x;
    uint8_t c = 0;
    if (reliable[i] && !finitef(hmap[i])) {
        const int r = ksize / 2;
        int known = 0, cont = 0;
        for (int dy = -r; dy <= r; dy++) {
            const size_t row = base + (size_t)reflect101(y + dy, h) * w;
            for (int dx = -r; dx <= r; dx++) {
                const size_t q = row + reflect101(x + dx, w);
                const int cq = reliable[q] != 0;
                cont += cq;
                known += cq && finitef(hmap[q]);
            }
        }
        const float frac = __fdiv_rn((float)known, __fadd_rn((float)cont, 1e-6f));      // float32 arrays upstream
        c = (uint8_t)(frac >= frac_thr && dist[i] >= min_dist);
    }
    cand[i] = c;
}
void launch_hole_candidates(const float *hmap, const uint8_t *reliable, const float *dist, int ksize, float frac_thr, float min_dist, uint8_t *cand,
                            int B, int h, int w, hipStream_t st)
{
    hipLaunchKernelGGL(k_hole_candidates, dim3((w + 63) / 64, h, B), dim3(64), 0, st, hmap, reliable, dist, ksize, frac_thr, min_dist, cand, h, w);
}

// tmp of :1787-1789 restricted to inpaint_only_mask's `known` (:1189): the known heights and `med` at the reliable NaN pixels that are not
// candidates; NaN elsewhere (so that the median over the finite values of this plane is fill_val, :1190)
__global__ void k_hole_tmp(const float *__restrict__ hmap, const uint8_t *__restrict__ reliable, const uint8_t *__restrict__ cand,
                           const float *__restrict__ med, float *__restrict__ tmp, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float v = hl_nan();
    if (reliable[i] && !cand[i]) {
        const float hv = hmap[i];
        float m = med[b];
        if (!finitef(m)) m = 0.f;                 // no known pixel at all: med = 0.0 (:1788)
        v = finitef(hv) ? hv : m;
    }
    tmp[i] = v;
}
// zin (:1192-1193): fill_val everywhere, the `known` values on top
__global__ void k_hole_zin(float *__restrict__ tmp_zin, const float *__restrict__ fill, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float f = fill[b];
    if (!finitef(f)) f = 0.f;                     // no known pixel: fill_val = 0.0 (:1190)
    const float v = tmp_zin[i];
    tmp_zin[i] = finitef(v) ? v : f;
}
void launch_hole_tmp(const float *hmap, const uint8_t *reliable, const uint8_t *cand, const float *med, float *tmp, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_hole_tmp, dim3((P + 255) / 256, B), dim3(256), 0, st, hmap, reliable, cand, med, tmp, P);
}
void launch_hole_zin(float *tmp_zin, const float *fill, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_hole_zin, dim3((P + 255) / 256, B), dim3(256), 0, st, tmp_zin, fill, P);
}

// height_rel_filled (:1773-1774, :1798) and output_reliable (:1801)
__global__ void k_hole_merge(float *__restrict__ hmap, const uint8_t *__restrict__ reliable, const uint8_t *__restrict__ cand,
                             const float *__restrict__ zin, uint8_t *__restrict__ out_rel, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float v = hmap[i];
    if (cand[i]) v = zin[i];
    const bool rel = reliable[i] != 0;
    if (!rel) v = hl_nan();
    hmap[i] = v;
    out_rel[i] = (uint8_t)(rel && finitef(v));
}
void launch_hole_merge(float *hmap, const uint8_t *reliable, const uint8_t *cand, const float *zin, uint8_t *out_rel, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_hole_merge, dim3((P + 255) / 256, B), dim3(256), 0, st, hmap, reliable, cand, zin, out_rel, P);
}

}  // namespace vf
