// Per-pixel / stencil kernels of the FTP preprocessing (bandwidth-bound; planes are [B, h, w]).
//   to_gray        cv2.cvtColor(BGR2GRAY) fixed point          (shape_ftp.py:1511-1512)
//   sobel_mag      cv2.Sobel x/y ksize 3 + magnitude            (shape_ftp.py:633-635)
//   bad_flags      (img >= hi) | (grad >= g) & valid            (shape_ftp.py:638-639)
//   morph          cv2.dilate / erode with ELLIPSE element      (shape_ftp.py:644-646, :758-760, :1734-1736)
//   gauss_rows/cols cv2.GaussianBlur((0,0), sigma) REFLECT_101  (shape_ftp.py:746, :831, :836, :1145-1146)
//   illum_norm     I/(blur+1e-6) - 1                            (shape_ftp.py:832)
#include "kernels.hpp"

namespace vf {

// ------------------------------------------------------------------------------------------------
__global__ void k_to_gray(const void *__restrict__ frames, int format, float *__restrict__ gray, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float v;
    if (format == 0) v = (float)((const uint8_t *)frames)[i];
    else if (format == 2) v = rintf(__half2float(((const __half *)frames)[i]));
    else {
        int bb, gg, rr;
        if (format == 1) { const uint8_t *s = (const uint8_t *)frames + 3 * i; bb = s[0]; gg = s[1]; rr = s[2]; }
        else { const __half *s = (const __half *)frames + 3 * i; bb = (int)rintf(__half2float(s[0])); gg = (int)rintf(__half2float(s[1])); rr = (int)rintf(__half2float(s[2])); }
        // OpenCV 4.x RGB2Gray 8u: descale(b*BY15 + g*GY15 + r*RY15, 15), BY15=3735 GY15=19235 RY15=9798
        v = (float)((bb * 3735 + gg * 19235 + rr * 9798 + (1 << 14)) >> 15);
    }
    gray[i] = v;
}

void launch_to_gray(const void *frames, int format, float *gray, int B, int P, hipStream_t st)
{
    dim3 grid((P + 255) / 256, B);
    hipLaunchKernelGGL(k_to_gray, grid, dim3(256), 0, st, frames, format, gray, P);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_sobel_mag(const float *__restrict__ img, float *__restrict__ grad, int h, int w)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const float *s = img + b * (size_t)h * w;
    int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h), xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    float a00 = s[(size_t)ym * w + xm], a01 = s[(size_t)ym * w + x], a02 = s[(size_t)ym * w + xp];
    float a10 = s[(size_t)y * w + xm], a12 = s[(size_t)y * w + xp];
    float a20 = s[(size_t)yp * w + xm], a21 = s[(size_t)yp * w + x], a22 = s[(size_t)yp * w + xp];
    float gx = __fadd_rn(__fadd_rn(__fsub_rn(a02, a00), __fmul_rn(2.0f, __fsub_rn(a12, a10))), __fsub_rn(a22, a20));
    float gy = __fadd_rn(__fadd_rn(__fsub_rn(a20, a00), __fmul_rn(2.0f, __fsub_rn(a21, a01))), __fsub_rn(a22, a02));
    grad[b * (size_t)h * w + (size_t)y * w + x] = sqrtf(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)));
}

void launch_sobel_mag(const float *img, float *grad, int B, int h, int w, hipStream_t st)
{
    dim3 grid((w + 255) / 256, h, B);
    hipLaunchKernelGGL(k_sobel_mag, grid, dim3(256), 0, st, img, grad, h, w);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_bad_flags(const float *__restrict__ img, const float *__restrict__ grad, const uint8_t *__restrict__ valid,
                            const float *__restrict__ thr_hi, const float *__restrict__ thr_g, uint8_t *__restrict__ bad, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    bad[i] = (uint8_t)(valid[p] && ((img[i] >= thr_hi[b]) || (grad[i] >= thr_g[b])));
}

void launch_bad_flags(const float *img, const float *grad, const uint8_t *valid, const float *thr_hi, const float *thr_g,
                      uint8_t *bad, int B, int P, hipStream_t st)
{
    dim3 grid((P + 255) / 256, B);
    hipLaunchKernelGGL(k_bad_flags, grid, dim3(256), 0, st, img, grad, valid, thr_hi, thr_g, bad, P);
}

// ------------------------------------------------------------------------------------------------
// Binary dilate / erode, element given as row spans; out-of-image pixels never contribute
// (cv::morphologyDefaultBorderValue).  Optional AND with a static [P] and/or a per-frame [B,P] mask.
__global__ void k_morph(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w, RowSpanSE se, int dilate,
                        const uint8_t *__restrict__ and_static, const uint8_t *__restrict__ and_frame)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const uint8_t *s = src + b * (size_t)h * w;
    int r = se.k / 2;
    // no early exit and clamped addresses: every load is independent of the others, so they are issued back to back
    // instead of one memory round trip per element pixel
    int any = 0, all = 1;
    for (int i = 0; i < se.k; i++) {
        const int yy = y + i - r;
        const bool rowin = yy >= 0 && yy < h;
        const uint8_t *row = s + (size_t)(yy < 0 ? 0 : (yy >= h ? h - 1 : yy)) * w;
        const int lo = se.lo[i], hi = se.hi[i];
#pragma unroll 4
        for (int dx = lo; dx <= hi; dx++) {
            const int xx = x + dx;
            const bool in = rowin && xx >= 0 && xx < w;
            const uint8_t px = row[xx < 0 ? 0 : (xx >= w ? w - 1 : xx)];
            any |= (in && px) ? 1 : 0;
            all &= (!in || px) ? 1 : 0;
        }
    }
    int v = dilate ? any : all;
    size_t p = (size_t)y * w + x;
    if (and_static && !and_static[p]) v = 0;
    if (and_frame && !and_frame[b * (size_t)h * w + p]) v = 0;
    dst[b * (size_t)h * w + p] = (uint8_t)v;
}

// Row-prefix variant for wide elements: inc[y][x] = number of set pixels in row y up to and including x, so
// "any / all set in [x0, x1]" is two loads per element row instead of a scan of the span.
__global__ __launch_bounds__(256) void k_row_prefix(const uint8_t *__restrict__ src, uint16_t *__restrict__ inc, int rows, int w)
{
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const uint8_t *s = src + (size_t)row * w;
    uint16_t *o = inc + (size_t)row * w;
    const unsigned long long le_mask = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    unsigned int run = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
        int x = x0 + lane;
        bool on = x < w && s[x] != 0;
        unsigned long long bm = __ballot(on);
        if (x < w) o[x] = (uint16_t)(run + (unsigned int)__popcll(bm & le_mask));
        run += (unsigned int)__popcll(bm);
    }
}

__global__ void k_morph_prefix(const uint16_t *__restrict__ inc, uint8_t *__restrict__ dst, int h, int w, RowSpanSE se, int dilate,
                               const uint8_t *__restrict__ and_static, const uint8_t *__restrict__ and_frame)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const uint16_t *I = inc + b * (size_t)h * w;
    int r = se.k / 2;
    // no early exit: the 2 * k loads are independent, so they are all in flight together (one memory round trip
    // per pixel instead of up to 2 * k dependent ones)
    int any = 0, all = 1;
#pragma unroll 8
    for (int i = 0; i < se.k; i++) {
        int yy = y + i - r;
        int x0 = x + se.lo[i], x1 = x + se.hi[i];
        if (x0 < 0) x0 = 0;
        if (x1 > w - 1) x1 = w - 1;
        const bool valid = yy >= 0 && yy < h && x1 >= x0;
        const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
        const int x1c = x1 < 0 ? 0 : x1, x0c = x0 > 0 ? x0 - 1 : 0;
        const uint16_t *row = I + (size_t)yc * w;
        const int hi = (int)row[x1c], lo = (int)row[x0c];
        const int cnt = hi - (x0 > 0 ? lo : 0);
        any |= (valid && cnt > 0) ? 1 : 0;
        all &= (!valid || cnt == x1 - x0 + 1) ? 1 : 0;
    }
    int v = dilate ? any : all;
    size_t p = (size_t)y * w + x;
    if (and_static && !and_static[p]) v = 0;
    if (and_frame && !and_frame[b * (size_t)h * w + p]) v = 0;
    dst[b * (size_t)h * w + p] = (uint8_t)v;
}

// Bit-plane variant for symmetric row spans [-a_i, a_i] (every cv ELLIPSE / RECT element): one workgroup per frame
// packs the mask into 64-bit words with ballots (coalesced byte loads), dilates in LDS with word shifts -- a
// 15x15 element costs a few hundred 64-bit ops per output word -- and unpacks.  Erosion is the dual: complement inside
// the image, dilate, complement (out-of-image pixels never erode: cv::morphologyDefaultBorderValue).
constexpr int MB_T = 1024;      // 16 waves: packing / unpacking a row is a global-memory round trip, so rows in flight are what counts
constexpr int MB_RB = 2;        // rows per wave and iteration (MB_RB * 4 independent byte loads in flight per lane)
constexpr int MB_MAXOPS = 4;
struct MorphSeq {               // up to MB_MAXOPS operations applied back to back on the packed plane (close = dilate, erode; n-fold dilate)
    int n;
    int dilate[MB_MAXOPS];
    RowSpanSE se[MB_MAXOPS];
};
__global__ __launch_bounds__(MB_T) void k_morph_bits(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w, MorphSeq seq,
                                                     const uint8_t *__restrict__ and_static, const uint8_t *__restrict__ and_frame)
{
    extern __shared__ unsigned long long mb_lds[];
    const int W64 = (w + 63) >> 6, nw = h * W64;
    unsigned long long *A = mb_lds, *O = mb_lds + nw;
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t *s = src + b * (size_t)h * w;
    // pack: the byte loads of MB_RB rows are independent and issued together
    for (int y0 = wid * MB_RB; y0 < h; y0 += (MB_T / 64) * MB_RB)
        for (int j0 = 0; j0 < W64; j0 += 4) {
            uint8_t px[MB_RB][4];
#pragma unroll
            for (int rr = 0; rr < MB_RB; rr++)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int y = y0 + rr, x = (j0 + u) * 64 + lane;
                    px[rr][u] = (y < h && j0 + u < W64 && x < w) ? s[(size_t)y * w + x] : (uint8_t)0;
                }
#pragma unroll
            for (int rr = 0; rr < MB_RB; rr++)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const unsigned long long word = __ballot(px[rr][u] != 0);
                    if (lane == 0 && y0 + rr < h && j0 + u < W64) A[(y0 + rr) * W64 + j0 + u] = word;
                }
        }
    __syncthreads();
    const unsigned long long last_valid = (w & 63) ? ((1ull << (w & 63)) - 1ull) : ~0ull;     // columns of the last word that exist
    for (int op = 0; op < seq.n; op++) {
        const RowSpanSE &se = seq.se[op];
        const bool dil = seq.dilate[op] != 0;
        if (!dil) {                                             // erosion = complement inside the image, dilate, complement
            for (int t = tid; t < nw; t += MB_T) { const int j = t % W64; A[t] = ~A[t] & (j == W64 - 1 ? last_valid : ~0ull); }
            __syncthreads();
        }
        const int r = se.k / 2;
        for (int t = tid; t < nw; t += MB_T) {
            const int y = t / W64, j = t - y * W64;
            unsigned long long acc = 0ull;
            for (int i = 0; i < se.k; i++) {
                const int yy = y + i - r, a = se.hi[i];
                if (yy < 0 || yy >= h || a < 0) continue;
                const unsigned long long *row = A + yy * W64;
                const unsigned long long wc = row[j], wl = j > 0 ? row[j - 1] : 0ull, wr = j < W64 - 1 ? row[j + 1] : 0ull;
                unsigned long long m = wc;
                for (int dd = 1; dd <= a; dd++) m |= (wc >> dd) | (wr << (64 - dd)) | (wc << dd) | (wl >> (64 - dd));
                acc |= m;
            }
            const unsigned long long valid = j == W64 - 1 ? last_valid : ~0ull;
            O[t] = dil ? (acc & valid) : (~acc & valid);
        }
        __syncthreads();
        unsigned long long *tmp = A; A = O; O = tmp;
    }
    // unpack (A holds the result); the optional AND masks of a row are loaded together
    uint8_t *o = dst + b * (size_t)h * w;
    for (int y0 = wid * MB_RB; y0 < h; y0 += (MB_T / 64) * MB_RB)
        for (int j0 = 0; j0 < W64; j0 += 4) {
            uint8_t ms[MB_RB][4], mf[MB_RB][4];
#pragma unroll
            for (int rr = 0; rr < MB_RB; rr++)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int y = y0 + rr, x = (j0 + u) * 64 + lane;
                    const size_t p = (y < h && j0 + u < W64 && x < w) ? (size_t)y * w + x : 0;
                    ms[rr][u] = and_static ? and_static[p] : (uint8_t)1;
                    mf[rr][u] = and_frame ? and_frame[b * (size_t)h * w + p] : (uint8_t)1;
                }
#pragma unroll
            for (int rr = 0; rr < MB_RB; rr++)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int y = y0 + rr, x = (j0 + u) * 64 + lane;
                    if (y >= h || j0 + u >= W64 || x >= w) continue;
                    const int v = (int)((A[y * W64 + j0 + u] >> lane) & 1ull);
                    o[(size_t)y * w + x] = (uint8_t)(v && ms[rr][u] && mf[rr][u]);
                }
        }
}

static bool morph_bits_ok(const RowSpanSE &se, int h, int w)
{
    for (int i = 0; i < se.k; i++) {
        if (se.lo[i] > se.hi[i]) { if (se.hi[i] >= 0) return false; continue; }      // empty rows must read hi < 0
        if (se.lo[i] != -se.hi[i] || se.hi[i] > 63) return false;
    }
    return (size_t)h * ((w + 63) >> 6) * 16 <= 64 * 1024;
}

void launch_morph(const uint8_t *src, uint8_t *dst, int B, int h, int w, const RowSpanSE &se, bool dilate,
                  const uint8_t *and_static, const uint8_t *and_frame, hipStream_t st, uint16_t *prefix_scratch)
{
    if (morph_bits_ok(se, h, w)) {
        MorphSeq seq;
        seq.n = 1; seq.dilate[0] = dilate ? 1 : 0; seq.se[0] = se;
        size_t lds = (size_t)h * ((w + 63) >> 6) * 16;
        hipLaunchKernelGGL(k_morph_bits, dim3(B), dim3(MB_T), lds, st, src, dst, h, w, seq, and_static, and_frame);
        return;
    }
    dim3 grid((w + 255) / 256, h, B);
    if (prefix_scratch && se.k >= 7 && w < 65536) {
        int rows = B * h;
        hipLaunchKernelGGL(k_row_prefix, dim3((rows + 3) / 4), dim3(256), 0, st, src, prefix_scratch, rows, w);
        hipLaunchKernelGGL(k_morph_prefix, grid, dim3(256), 0, st, prefix_scratch, dst, h, w, se, dilate ? 1 : 0, and_static, and_frame);
        return;
    }
    hipLaunchKernelGGL(k_morph, grid, dim3(256), 0, st, src, dst, h, w, se, dilate ? 1 : 0, and_static, and_frame);
}

// n operations with the same element back to back (dilates[i] != 0: dilate, else erode); the AND masks apply to the final result.
// One kernel (pack once, unpack once) when the bit-plane path applies; `tmp` is a scratch plane for the fallback chain.
void launch_morph_seq(const uint8_t *src, uint8_t *dst, uint8_t *tmp, int B, int h, int w, const RowSpanSE &se, const int *dilates, int n,
                      const uint8_t *and_static, const uint8_t *and_frame, hipStream_t st, uint16_t *prefix_scratch)
{
    if (n >= 1 && n <= MB_MAXOPS && morph_bits_ok(se, h, w)) {
        MorphSeq seq;
        seq.n = n;
        for (int i = 0; i < n; i++) { seq.dilate[i] = dilates[i] ? 1 : 0; seq.se[i] = se; }
        size_t lds = (size_t)h * ((w + 63) >> 6) * 16;
        hipLaunchKernelGGL(k_morph_bits, dim3(B), dim3(MB_T), lds, st, src, dst, h, w, seq, and_static, and_frame);
        return;
    }
    const uint8_t *cur = src;
    for (int i = 0; i < n; i++) {
        // alternate so that the last operation writes dst
        uint8_t *out = ((n - 1 - i) & 1) ? tmp : dst;
        const bool last = i == n - 1;
        launch_morph(cur, out, B, h, w, se, dilates[i] != 0, last ? and_static : nullptr, last ? and_frame : nullptr, st, prefix_scratch);
        cur = out;
    }
}

// ------------------------------------------------------------------------------------------------
// Separable Gaussian, BORDER_REFLECT_101.  Row pass: 64-column x 16-row tile staged in LDS with halo.
constexpr int GB_TX = 64, GB_TY = 16, GB_MAXK = 512;

__global__ __launch_bounds__(256) void k_gauss_rows(const float *__restrict__ src, float *__restrict__ dst,
                                                    const float *__restrict__ kern, int ksize, int h, int w)
{
    // 64 columns x GB_TY rows per workgroup; a thread produces the same column of GB_TY / 4 rows, so a tap fetched once (a scalar
    // load: wave-uniform address in read-only memory) feeds that many outputs and the row loads of the tile are all in flight together
    extern __shared__ float lds[];
    constexpr int NR = GB_TY / 4;
    const int r = ksize / 2;
    const int tw = GB_TX + 2 * r;
    float *tile = lds;                  // GB_TY rows of tw
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x0 = blockIdx.x * GB_TX, y0 = blockIdx.y * GB_TY;
    const size_t b = blockIdx.z;
    const float *plane = src + b * (size_t)h * w;
    for (int rr = ty; rr < GB_TY; rr += 4) {
        const int y = y0 + rr;
        if (y >= h) break;
        const float *row = plane + (size_t)y * w;
        for (int i = tx; i < tw; i += 64) tile[rr * tw + i] = row[reflect101(x0 + i - r, w)];
    }
    __syncthreads();
    const int x = x0 + tx;
    if (x >= w) return;
    float acc[NR];
    const float *t = tile + ty * tw + tx;
    {
        const float k0 = kern[0];
#pragma unroll
        for (int q = 0; q < NR; q++) acc[q] = k0 * t[q * 4 * tw];
    }
    for (int j = 1; j < ksize; j++) {
        const float kj = kern[j];
#pragma unroll
        for (int q = 0; q < NR; q++) acc[q] = fmaf(kj, t[q * 4 * tw + j], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int y = y0 + ty + 4 * q;
        if (y < h) dst[b * (size_t)h * w + (size_t)y * w + x] = acc[q];
    }
}

// Column pass: cv::SymmColumnFilter's symmetric form, s = k[r]*S[y]; s = fma(k[r+j], S[y+j] + S[y-j], s) for j = 1..r (the order the
// parity tests' CPU restatement executes too: same operations in the same order, so the same bits).  Each thread produces
// GC_R consecutive rows of one column; lanes run along x so every row read is coalesced.  The two source rows a step needs for its GC_R
// outputs are the previous step's shifted by one row, so a step costs two new reads (register windows `up` / `dn`).
constexpr int GC_R = 8;

template <class Load>
__device__ inline void gauss_col_symm(Load ld, const float *__restrict__ kern, int r, float (&acc)[GC_R])
{
    // ld(i): source value i rows below the centre of output 0 (i in [-r, GC_R - 1 + r])
    float up[GC_R], dn[GC_R];
    const float kc = kern[r];
#pragma unroll
    for (int o = 0; o < GC_R; o++) { const float v = ld(o); acc[o] = __fmul_rn(kc, v); up[o] = v; dn[o] = v; }
    for (int j = 1; j <= r; j++) {
        const float nu = ld(GC_R - 1 + j), nd = ld(-j);
#pragma unroll
        for (int o = 0; o < GC_R - 1; o++) up[o] = up[o + 1];
        up[GC_R - 1] = nu;
#pragma unroll
        for (int o = GC_R - 1; o > 0; o--) dn[o] = dn[o - 1];
        dn[0] = nd;
        const float kj = kern[r + j];
#pragma unroll
        for (int o = 0; o < GC_R; o++) acc[o] = fmaf(kj, __fadd_rn(up[o], dn[o]), acc[o]);
    }
}

// global-memory form (kernels too long for the LDS tile)
__global__ __launch_bounds__(256) void k_gauss_cols(const float *__restrict__ src, float *__restrict__ dst,
                                                    const float *__restrict__ kern, int ksize, int h, int w)
{
    const int r = ksize / 2;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * GC_R;
    const size_t b = blockIdx.z;
    if (x >= w || y0 >= h) return;
    const float *s = src + b * (size_t)h * w;
    float acc[GC_R];
    gauss_col_symm([&](int i) { return s[(size_t)reflect101(y0 + i, h) * w + x]; }, kern, r, acc);
#pragma unroll
    for (int o = 0; o < GC_R; o++)
        if (y0 + o < h) dst[b * (size_t)h * w + (size_t)(y0 + o) * w + x] = acc[o];
}

// Column pass through LDS: the block's 64 columns x (32 + 2r) source rows are staged with ONE round of independent coalesced loads
// (the global form walks its taps with dependent loads), then every thread runs the symmetric sum over its 8 output rows.
__global__ __launch_bounds__(256) void k_gauss_cols_lds(const float *__restrict__ src, float *__restrict__ dst,
                                                        const float *__restrict__ kern, int ksize, int h, int w)
{
    extern __shared__ float lds[];
    float *tile = lds;                     // [(32 + 2r)][64]
    const int r = ksize / 2, rows = 32 + 2 * r;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + tx;
    const int yb = blockIdx.y * 32;
    const size_t b = blockIdx.z;
    const float *s = src + b * (size_t)h * w;
    const int xc = x < w ? x : w - 1;
    for (int j = ty; j < rows; j += 4) tile[j * 64 + tx] = s[(size_t)reflect101(yb + j - r, h) * w + xc];
    __syncthreads();
    const int y0 = yb + ty * GC_R;
    if (x >= w || y0 >= h) return;
    const float *tc = tile + (ty * GC_R + r) * 64 + tx;
    float acc[GC_R];
    gauss_col_symm([&](int i) { return tc[i * 64]; }, kern, r, acc);
#pragma unroll
    for (int o = 0; o < GC_R; o++)
        if (y0 + o < h) dst[b * (size_t)h * w + (size_t)(y0 + o) * w + x] = acc[o];
}

void launch_gauss_rows(const float *src, float *dst, const float *kern, int ksize, int B, int h, int w, hipStream_t st)
{
    dim3 grid((w + GB_TX - 1) / GB_TX, (h + GB_TY - 1) / GB_TY, B);
    size_t lds = (size_t)GB_TY * (GB_TX + 2 * (ksize / 2)) * sizeof(float);
    hipLaunchKernelGGL(k_gauss_rows, grid, dim3(256), lds, st, src, dst, kern, ksize, h, w);
}

void launch_gauss_cols(const float *src, float *dst, const float *kern, int ksize, int B, int h, int w, hipStream_t st)
{
    dim3 grid((w + 63) / 64, (h + 4 * GC_R - 1) / (4 * GC_R), B);
    const size_t lds = (size_t)(32 + 2 * (ksize / 2)) * 64 * sizeof(float);
    if (lds <= 64 * 1024) { hipLaunchKernelGGL(k_gauss_cols_lds, grid, dim3(256), lds, st, src, dst, kern, ksize, h, w); return; }
    hipLaunchKernelGGL(k_gauss_cols, grid, dim3(256), 0, st, src, dst, kern, ksize, h, w);
}

// Both passes in one kernel for short kernels (ksize <= GF_MAXK): a 64 x 32 output tile with its halo goes through LDS once -- row pass
// into a second LDS plane (rounded to float exactly as the intermediate plane of the two-kernel path is), column pass out of it -- so the
// intermediate plane never travels to memory.  Same taps, same order of the operations per output: same bits as the two kernels.
constexpr int GF_TX = 64, GF_TY = 32, GF_MAXK = 15;
__global__ __launch_bounds__(256) void k_gauss_fused(const float *__restrict__ src, float *__restrict__ dst, const float *__restrict__ kern,
                                                     int ksize, int h, int w)
{
    __shared__ float in_t[(GF_TY + GF_MAXK - 1) * (GF_TX + GF_MAXK - 1)];
    __shared__ float mid_t[(GF_TY + GF_MAXK - 1) * GF_TX];
    const int r = ksize / 2;
    const int tw = GF_TX + 2 * r, th = GF_TY + 2 * r;
    const int x0 = blockIdx.x * GF_TX, y0 = blockIdx.y * GF_TY;
    const size_t b = blockIdx.z;
    const float *plane = src + b * (size_t)h * w;
    // tile with halo: a wave per tile row, the lane's (at most two) source columns reflected once
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int sx0 = reflect101(x0 + lane - r, w), sx1 = reflect101(x0 + lane + 64 - r, w);
        const bool two = lane + 64 < tw;
        for (int ty = wv; ty < th; ty += 4) {
            const float *row = plane + (size_t)reflect101(y0 + ty - r, h) * w;
            in_t[ty * tw + lane] = row[sx0];
            if (two) in_t[ty * tw + lane + 64] = row[sx1];
        }
    }
    __syncthreads();
    // row pass: th rows x 64 columns, FOUR adjacent outputs per thread out of one sliding window of taps (ksize + 3 LDS reads instead of
    // 4 ksize); every output is still  s = k[0] S[0]; s = fma(k[j], S[j], s)  in ascending j
    for (int q = threadIdx.x; q < th * 16; q += 256) {
        const int ty = q >> 4, tx = (q & 15) * 4;
        const float *t = in_t + ty * tw + tx;
        float v0 = t[0], v1 = t[1], v2 = t[2], v3 = t[3];
        const float k0 = kern[0];
        float a0 = k0 * v0, a1 = k0 * v1, a2 = k0 * v2, a3 = k0 * v3;
        for (int j = 1; j < ksize; j++) {
            v0 = v1; v1 = v2; v2 = v3; v3 = t[j + 3];
            const float kj = kern[j];
            a0 = fmaf(kj, v0, a0); a1 = fmaf(kj, v1, a1); a2 = fmaf(kj, v2, a2); a3 = fmaf(kj, v3, a3);
        }
        float *m = mid_t + ty * GF_TX + tx;
        m[0] = a0; m[1] = a1; m[2] = a2; m[3] = a3;
    }
    __syncthreads();
    // column pass: symmetric sum over GC_R output rows of one column (gauss_col_symm)
    const int tx = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int x = x0 + tx, yb = y0 + tg * GC_R;
    if (x >= w || yb >= h) return;
    const float *tc = mid_t + (tg * GC_R + r) * GF_TX + tx;
    float acc[GC_R];
    gauss_col_symm([&](int i) { return tc[i * GF_TX]; }, kern, r, acc);
#pragma unroll
    for (int o = 0; o < GC_R; o++)
        if (yb + o < h) dst[b * (size_t)h * w + (size_t)(yb + o) * w + x] = acc[o];
}

// separable blur src -> dst (tmp: intermediate plane of the two-kernel path)
void launch_gauss_blur(const float *src, float *tmp, float *dst, const float *kern, int ksize, int B, int h, int w, hipStream_t st)
{
    static_assert(GF_TY == 4 * GC_R, "four waves of GC_R rows");
    if (ksize <= GF_MAXK && src != dst) {
        hipLaunchKernelGGL(k_gauss_fused, dim3((w + GF_TX - 1) / GF_TX, (h + GF_TY - 1) / GF_TY, B), dim3(256), 0, st, src, dst, kern, ksize, h, w);
        return;
    }
    launch_gauss_rows(src, tmp, kern, ksize, B, h, w, st);
    launch_gauss_cols(tmp, dst, kern, ksize, B, h, w, st);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_illum_norm(const float *__restrict__ img, const float *__restrict__ blur, float *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // float32: img / (blur + 1e-6) - 1.0
    out[i] = __fsub_rn(__fdiv_rn(img[i], __fadd_rn(blur[i], 1e-6f)), 1.0f);
}
void launch_illum_norm(const float *img, const float *blur, float *out, int B, int P, hipStream_t st)
{
    size_t n = (size_t)B * P;
    hipLaunchKernelGGL(k_illum_norm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, img, blur, out, n);
}

__global__ void k_mul_static(const float *__restrict__ a, const float *__restrict__ stat, float *__restrict__ out, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    out[b * (size_t)P + p] = __fmul_rn(a[b * (size_t)P + p], stat[p]);
}
void launch_mul_static(const float *a, const float *stat, float *out, int B, int P, hipStream_t st)
{
    dim3 grid((P + 255) / 256, B);
    hipLaunchKernelGGL(k_mul_static, grid, dim3(256), 0, st, a, stat, out, P);
}

__global__ void k_count_u8(const uint8_t *__restrict__ m, int *__restrict__ counts, int P)
{
    __shared__ int scratch[16];
    size_t b = blockIdx.y;
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) c += m[b * (size_t)P + p] != 0;
    c = block_sum<int>(c, scratch);
    if (threadIdx.x == 0 && c) atomicAdd(&counts[b], c);
}
void launch_count_u8(const uint8_t *m, int *counts, int B, int P, hipStream_t st)
{
    hipMemsetAsync(counts, 0, sizeof(int) * B, st);
    int gx = (P + 256 * 16 - 1) / (256 * 16);
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_count_u8, dim3(gx, B), dim3(256), 0, st, m, counts, P);
}

}  // namespace vf
