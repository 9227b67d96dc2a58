// cv2.inpaint(INPAINT_TELEA) -- cluster bookkeeping for the cluster-parallel march (shape_ftp.py:652-666).
//
// Telea's march is ordered by a stable priority queue (T, push sequence), but hole pixels farther apart
// than their (2*range+1)^2 neighbourhoods plus the outside T ring can never see each other: restricted to
// such an independent CLUSTER of hole pixels the queue pops in exactly the same relative order.  So:
//   1. dilate the hole mask by a (2*(range+1)+1)^2 square and label its 8-connected components
//      (k_morph_bits + k_cc_label): hole pixels of different components are > 2*range+3 apart (Chebyshev),
//      while rings / neighbourhoods only reach across gaps of at most 2*range+1;
//   2. k_cluster_bbox / k_cluster_list: bounding box of the hole pixels of every component and the list of
//      components per frame; components whose window (bbox grown by range+1) exceeds the cluster kernel's
//      capacity are flagged `big`;
//   3. k_telea_clusters2 (k_inpaint_win.hip) marches every listed cluster on its own LDS window, one wave each,
//      and flags the ones whose queue overflowed;
//   4. k_split_bad builds the hole mask of the flagged clusters for the frame-window / whole-frame kernels.
#include "kernels.hpp"

namespace vf {

// box planes (int32 [B,P], indexed by component root): xmin / ymin start at 0x7f7f7f7f, xmax / ymax at 0
__global__ void k_cluster_bbox(const uint8_t *__restrict__ bad, const int32_t *__restrict__ labels, int32_t *xmin, int32_t *ymin,
                               int32_t *xmax, int32_t *ymax, int h, int w)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    int P = h * w;
    const bool on = p < P && bad[b * (size_t)P + (p < P ? p : 0)] != 0;
    const int lab = on ? labels[b * (size_t)P + p] : -1;
    int y = p / w, x = p - y * w;
    // the hole pixels of a wave (64 consecutive pixels) mostly belong to one cluster: one set of atomics per cluster and wave instead of one
    // per pixel (the thousands of pixels of a big cluster all hit the same four words)
    unsigned long long active = __ballot(on);
    while (active) {
        const int leader = __ffsll((long long)active) - 1;
        const int r0 = __builtin_amdgcn_readlane(lab, leader);
        const unsigned long long same = __ballot(on && lab == r0);
        const bool mine = on && lab == r0;
        int x0 = mine ? x : 0x7fffffff, y0 = mine ? y : 0x7fffffff, x1 = mine ? x : -1, y1 = mine ? y : -1;
        for (int o = 32; o; o >>= 1) {
            x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
            x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
        }
        if ((int)(threadIdx.x & 63) == leader) {
            const size_t r = b * (size_t)P + r0;
            atomicMin(&xmin[r], x0); atomicMin(&ymin[r], y0); atomicMax(&xmax[r], x1); atomicMax(&ymax[r], y1);
        }
        active &= ~same;
    }
}

// list[b*P + k] = root of the k-th cluster of frame b that fits the cluster kernel; big[root] = 1 for the others
__global__ void k_cluster_list(const int32_t *__restrict__ labels, const int32_t *__restrict__ xmin, const int32_t *__restrict__ ymin,
                               const int32_t *__restrict__ xmax, const int32_t *__restrict__ ymax, int32_t *__restrict__ list,
                               int32_t *__restrict__ count, uint8_t *__restrict__ big, int range, int cells_cap, int h, int w)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    int P = h * w;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    big[i] = 0;
    if (labels[i] != p || xmin[i] == 0x7f7f7f7f) return;
    const int M = range + 1;
    const int cells = (ymax[i] - ymin[i] + 1 + 2 * M) * (xmax[i] - xmin[i] + 1 + 2 * M);     // unclipped window (k_inpaint_win.hip)
    if (cells > cells_cap) big[i] = 1;
    else { int k = atomicAdd(&count[b], 1); list[b * (size_t)P + k] = p; }
}

// after the LDS march: the roots still flagged `big` (window too large, or the queue of their window overflowed), compacted into `list`
__global__ void k_cluster_list_big(const int32_t *__restrict__ labels, const int32_t *__restrict__ xmin, const uint8_t *__restrict__ big,
                                   int32_t *__restrict__ list, int32_t *__restrict__ count, int h, int w)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    int P = h * w;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    if (labels[i] != p || xmin[i] == 0x7f7f7f7f || !big[i]) return;
    int k = atomicAdd(&count[b], 1);
    list[b * (size_t)P + k] = p;
}

// bad_big = bad & big[root]
__global__ void k_split_bad(const uint8_t *__restrict__ bad, const int32_t *__restrict__ labels, const uint8_t *__restrict__ big,
                            uint8_t *__restrict__ bad_big, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    bad_big[i] = (uint8_t)(bad[i] && big[b * (size_t)P + labels[i]]);
}

// scratch layout per frame: dil u8[P] | labels i32[P] | xmin,ymin,xmax,ymax i32[P] x4 | list i32[P] | big u8[P] | bad_big u8[P] | count
size_t inpaint_cl_scratch_bytes_per_frame(int h, int w)
{
    size_t P = (size_t)h * w;
    return P * (1 + 4 + 16 + 4 + 1 + 1) + 64;
}

bool inpaint_clusters_supported(int range) { return inpaint_big_supported(range); }     // the left-over clusters need k_inpaint_big.hip

int inpaint_cluster_cells_cap();
void launch_telea_clusters2(float *img, const uint8_t *bad, const int32_t *labels, const int32_t *list, const int32_t *count, const int32_t *xmin,
                            const int32_t *ymin, const int32_t *xmax, const int32_t *ymax, uint8_t *big, int range, int B, int h, int w, hipStream_t st);

// Marches every cluster that fits on its own window; *bad_big_out = hole mask of the clusters that were left over, *left = their roots
// (list / count per frame, in no particular order), labels and bounding boxes for launch_inpaint_big_clusters.
void launch_inpaint_clusters(float *img, const uint8_t *bad, int range, void *scratch, uint8_t **bad_big_out, ClusterPlanes *left, int B, int h, int w,
                             hipStream_t st)
{
    const int P = h * w;
    const size_t n = (size_t)B * P;
    uint8_t *base = (uint8_t *)scratch;
    uint8_t *dil = base; base += (n + 255) & ~(size_t)255;
    int32_t *labels = (int32_t *)base; base += n * 4;
    int32_t *xmin = (int32_t *)base; base += n * 4;
    int32_t *ymin = (int32_t *)base; base += n * 4;
    int32_t *xmax = (int32_t *)base; base += n * 4;
    int32_t *ymax = (int32_t *)base; base += n * 4;
    int32_t *list = (int32_t *)base; base += n * 4;
    uint8_t *big = base; base += (n + 255) & ~(size_t)255;
    uint8_t *bad_big = base; base += (n + 255) & ~(size_t)255;
    int32_t *count = (int32_t *)base;
    RowSpanSE se;
    const int R = range + 1;   // hole pixels interact only within Chebyshev distance 2*range+1; R = range would already separate them
    se.k = 2 * R + 1;
    for (int i = 0; i < se.k; i++) { se.lo[i] = (int8_t)(-R); se.hi[i] = (int8_t)R; }
    launch_morph(bad, dil, B, h, w, se, true, nullptr, nullptr, st);
    launch_cc_label(dil, labels, B, h, w, st);
    (void)hipMemsetAsync(xmin, 0x7f, n * 8, st);          // xmin, ymin
    (void)hipMemsetAsync(xmax, 0, n * 8, st);             // xmax, ymax
    (void)hipMemsetAsync(count, 0, (size_t)B * 4, st);
    dim3 g((P + 255) / 256, B);
    hipLaunchKernelGGL(k_cluster_bbox, g, dim3(256), 0, st, bad, labels, xmin, ymin, xmax, ymax, h, w);
    hipLaunchKernelGGL(k_cluster_list, g, dim3(256), 0, st, labels, xmin, ymin, xmax, ymax, list, count, big, range, inpaint_cluster_cells_cap(), h, w);
    launch_telea_clusters2(img, bad, labels, list, count, xmin, ymin, xmax, ymax, big, range, B, h, w, st);
    hipLaunchKernelGGL(k_split_bad, g, dim3(256), 0, st, bad, labels, big, bad_big, P);
    *bad_big_out = bad_big;
    if (left) {
        // the list of the clusters marched above is dead: it now takes the roots that are left
        (void)hipMemsetAsync(count, 0, (size_t)B * 4, st);
        hipLaunchKernelGGL(k_cluster_list_big, g, dim3(256), 0, st, labels, xmin, big, list, count, h, w);
        left->dil = dil; left->labels = labels; left->list = list; left->count = count; left->xmin = xmin; left->ymin = ymin; left->xmax = xmax; left->ymax = ymax;
    }
}

}  // namespace vf
