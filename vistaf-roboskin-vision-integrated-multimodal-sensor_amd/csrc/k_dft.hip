// Side-band demodulation as a pruned DFT (shape_ftp.py:857-997).
//
// The reference reflect-pads the n x n crop to N x N, takes fft2, keeps a (2*bw+1)^2 patch around the
// (locked) carrier, windows it, moves it to DC, takes ifft2, applies a sub-bin ramp and crops.  Only
// ph*pw (<= 21x21) spectral bins are ever used, so both transforms are computed directly:
//   forward  patch = win . ( Ey[ph x h] . (iw - mu)[h x w] . Ex[w x pw] )       (reflect pad folded into Ex/Ey)
//   inverse  field = Gy[h x ph] . patch[ph x pw] . Gx[pw x w]                   (crop, ramp, 1/N^2 folded in)
// Everything between the float32 input plane and the float32 results (amplitude, wrapped phase difference) is
// float64: float64 twiddles built on the host, float64 fused multiply-adds, float64 patch / field.  That is the
// arithmetic of the reference under NumPy < 2 (fft2 of a float32 array runs in complex128; the sub-bin ramp and
// cdef * conj(cref) are complex128 under every NumPy), to ~1e-15 relative -- so amplitude and wrapped phase round
// to the same float32 as a complex128 NumPy transform's (what the parity tests compare with) and every threshold downstream of them
// (quality >= p25) sees the same plane.  Tables may be per frame (tab_stride != 0: the uncached-pair mode, where
// every sample carries its own carrier) or shared by the batch (0).
#include <algorithm>
#include "kernels.hpp"

namespace vf {

constexpr int DFT_RB = 12;   // row groups per workgroup in stage 1 (12 * 21 = 252 threads)
constexpr int DFT_RR = 4;    // rows per thread in stage 1: every twiddle loaded feeds DFT_RR rows

__device__ inline void cmac(double &ar, double &ai, double er, double ei, double vr, double vi)
{
    ar = fma(er, vr, ar); ar = fma(-ei, vi, ar);
    ai = fma(er, vi, ai); ai = fma(ei, vr, ai);
}

// stage 1: T[b, y, c] = sum_x (iw[b,y,x] - mu[b]) * Ex[x, c].  Thread = (row group, c); the rows of the block are staged in LDS.
__global__ __launch_bounds__(256) void k_dft_fwd1(const float *__restrict__ iw, const float *__restrict__ mu,
                                                  const double2 *__restrict__ Ex_all, size_t ex_stride, double2 *__restrict__ T, int h, int w,
                                                  int pw, int rb)
{
    extern __shared__ float rows[];   // rb * DFT_RR * w
    size_t b = blockIdx.y;
    const double2 *Ex = Ex_all + b * ex_stride;
    int y0 = blockIdx.x * rb * DFT_RR;
    int nr = min(rb * DFT_RR, h - y0);
    const float *src = iw + b * (size_t)h * w + (size_t)y0 * w;
    float m = mu ? mu[b] : 0.f;
    for (int i = threadIdx.x; i < nr * w; i += blockDim.x) rows[i] = __fsub_rn(src[i], m);
    for (int i = nr * w + threadIdx.x; i < rb * DFT_RR * w; i += blockDim.x) rows[i] = 0.f;
    __syncthreads();
    int rg = threadIdx.x / pw, c = threadIdx.x % pw;
    if (rg >= rb || rg * DFT_RR >= nr) return;
    const float *r = rows + rg * DFT_RR * w;
    double ar[DFT_RR], ai[DFT_RR];
#pragma unroll
    for (int k = 0; k < DFT_RR; k++) { ar[k] = 0.0; ai[k] = 0.0; }
    for (int x = 0; x < w; x++) {
        const double2 e = Ex[(size_t)x * pw + c];
#pragma unroll
        for (int k = 0; k < DFT_RR; k++) {
            const double v = r[k * w + x];
            ar[k] = fma(v, e.x, ar[k]);
            ai[k] = fma(v, e.y, ai[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < DFT_RR; k++)
        if (rg * DFT_RR + k < nr) T[(b * (size_t)h + y0 + rg * DFT_RR + k) * pw + c] = make_double2(ar[k], ai[k]);
}

// stage 1 as a float64 GEMM on the matrix cores: C[row = (b, y)][n] = sum_x V[row][x] * E[x][n], n = 0..2*pw-1 (real parts of the pw bins,
// then their imaginary parts), with v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4] = V, B[k = lane >> 4][j = lane & 15] = E,
// D[row = (lane >> 4) + 4 * reg][col = lane & 15].  One wave owns a strip of 16 image rows and all column tiles (<= DFT_MAXT at a time);
// operands come straight from global memory / L2 (the table is 2 * pw * w doubles), one element per lane and k-step.  This is the only dense
// contraction of the path; float64 keeps the accumulation the parity of the demodulated field relies on.
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int DFT_MAXT = 4;            // column tiles (16 columns each) accumulated per pass over x
__global__ __launch_bounds__(256) void k_dft_fwd1_mfma(const float *__restrict__ iw, const float *__restrict__ mu, const double2 *__restrict__ Ex_all,
                                                       size_t ex_stride, double2 *__restrict__ T, int h, int w, int pw, int rows_total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wid) * 16;
    if (row0 >= rows_total) return;
    const int r = lane & 15, kk = lane >> 4;
    const int grow = min(row0 + r, rows_total - 1);            // rows past the end repeat the last one (never stored)
    const size_t b = (size_t)(row0 / h);                       // a strip never straddles frames in the table choice: h % 16 == 0 is required
    const double *E = (const double *)(Ex_all + b * ex_stride);
    const float *src = iw + (size_t)grow * w;
    const float m = mu ? mu[grow / h] : 0.f;
    const int ncol = 2 * pw, ntile = (ncol + 15) / 16;
    for (int t0 = 0; t0 < ntile; t0 += DFT_MAXT) {
        v4f64 acc[DFT_MAXT];
        int eoff[DFT_MAXT];                                     // offset of this lane's column inside one table row (doubles), -1: padding column
#pragma unroll
        for (int t = 0; t < DFT_MAXT; t++) {
            acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
            const int n = (t0 + t) * 16 + r;
            eoff[t] = (t0 + t < ntile && n < ncol) ? (n < pw ? 2 * n : 2 * (n - pw) + 1) : -1;
        }
        for (int x0 = 0; x0 < w; x0 += 4) {
            const int x = x0 + kk;
            const bool in = x < w;
            const double a = in ? (double)__fsub_rn(src[in ? x : 0], m) : 0.0;
            double bv[DFT_MAXT];
#pragma unroll
            for (int t = 0; t < DFT_MAXT; t++) bv[t] = (in && eoff[t] >= 0) ? E[(size_t)x * (2 * pw) + eoff[t]] : 0.0;
#pragma unroll
            for (int t = 0; t < DFT_MAXT; t++)
                if (t0 + t < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[t], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < DFT_MAXT; t++) {
            if (eoff[t] < 0) continue;
            const int n = (t0 + t) * 16 + r, c = n < pw ? n : n - pw;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = row0 + kk + 4 * q;
                if (row < rows_total) ((double *)(T + (size_t)row * pw + c))[n < pw ? 0 : 1] = acc[t][q];
            }
        }
    }
}

// stage 2: patch[b, a, c] = win[a,c] * sum_y Ey[a, y] * T[b, y, c]   (complex128 spectrum value times the float32 window, as upstream's
// `patch *= win` under complex128: both parts times the window in float64)
__global__ void k_dft_fwd2(const double2 *__restrict__ T, const double2 *__restrict__ Ey_all, size_t ey_stride, const float *__restrict__ win,
                           double2 *__restrict__ patch, int h, int ph, int pw, int pstride)
{
    size_t b = blockIdx.x;
    const double2 *Ey = Ey_all + b * ey_stride;
    const double2 *Tb = T + b * (size_t)h * pw;
    for (int t = threadIdx.x; t < ph * pw; t += blockDim.x) {       // one pass for patches up to 32 x 32 bins
        int a = t / pw, c = t % pw;
        double ar = 0.0, ai = 0.0;
        // one workgroup per frame: the loads of eight rows are issued together (the loop was one memory round trip per row)
        int y = 0;
        for (; y + 8 <= h; y += 8) {
            double2 e[8], v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { e[k] = Ey[(size_t)a * h + y + k]; v[k] = Tb[(size_t)(y + k) * pw + c]; }
#pragma unroll
            for (int k = 0; k < 8; k++) cmac(ar, ai, e[k].x, e[k].y, v[k].x, v[k].y);
        }
        for (; y < h; y++) {
            const double2 e = Ey[(size_t)a * h + y], v = Tb[(size_t)y * pw + c];
            cmac(ar, ai, e.x, e.y, v.x, v.y);
        }
        const double wv = (double)win[t];
        patch[b * (size_t)pstride + t] = make_double2(ar * wv, ai * wv);
    }
}

void launch_dft_forward(const float *iw, const float *mu, const double2 *Ex, const double2 *Ey, size_t tab_stride_x, size_t tab_stride_y,
                        const float *win, double2 *tmpT, double2 *patch, int patch_stride, int B, int h, int w, int ph, int pw, hipStream_t st)
{
    int rb = 256 / pw;
    if (rb > DFT_RB) rb = DFT_RB;
    const int rb_lds = (int)((60 * 1024) / ((size_t)DFT_RR * w * sizeof(float)));     // staged rows must fit the default dynamic LDS limit
    if (rb > rb_lds) rb = rb_lds;
    if (rb < 1) rb = 1;
    if (h % 16 == 0 || tab_stride_x == 0) {      // matrix-core form: a strip of 16 rows must use ONE table (shared, or inside one frame)
        const int rows = B * h;
        hipLaunchKernelGGL(k_dft_fwd1_mfma, dim3(((rows + 15) / 16 + 3) / 4), dim3(256), 0, st, iw, mu, Ex, tab_stride_x, tmpT, h, w, pw, rows);
    } else {
        dim3 g1((h + rb * DFT_RR - 1) / (rb * DFT_RR), B);
        hipLaunchKernelGGL(k_dft_fwd1, g1, dim3(256), (size_t)rb * DFT_RR * w * sizeof(float), st, iw, mu, Ex, tab_stride_x, tmpT, h, w, pw, rb);
    }
    hipLaunchKernelGGL(k_dft_fwd2, dim3(B), dim3(std::min(1024, ((ph * pw + 63) / 64) * 64)), 0, st, (const double2 *)tmpT, Ey, tab_stride_y, win, patch, h, ph, pw,
                       patch_stride);
}

// stage 3: Q[b, a, x] = sum_c patch[b, a, c] * Gx[c, x]
__global__ void k_dft_inv1(const double2 *__restrict__ patch, int pstride, const double2 *__restrict__ Gx_all, size_t gx_stride,
                           double2 *__restrict__ Q, int w, int ph, int pw)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int a = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const double2 *Gx = Gx_all + b * gx_stride;
    const double2 *pr = patch + b * (size_t)pstride + (size_t)a * pw;
    double ar = 0.0, ai = 0.0;
    for (int c = 0; c < pw; c++) {
        const double2 p = pr[c], g = Gx[(size_t)c * w + x];
        cmac(ar, ai, p.x, p.y, g.x, g.y);
    }
    Q[(b * (size_t)ph + a) * w + x] = make_double2(ar, ai);
}

// stage 4: field[b, y, x] = sum_a Gy[y, a] * Q[b, a, x]; amp = |field| (np.abs -> float32, :997); with a reference field (cref != null: the
// deformed frames) the same thread goes on to the phase difference angle(cdef * conj(cref)) -> float32 (:1681-1689) and the amplitude
// product amp_ref * amp_def (:742), so the float64 field never travels to memory unless `field` is given (reference frame, debug planes).
// On the matrix cores: the complex product field = Gy . Q is the real float64 GEMM  [re | im] = [Gr | Gi] . [[Qr, Qi], [-Qi, Qr]]
// with K = 2 * ph: A[y][k] = Gr[y][k] for k < ph, Gi[y][k - ph] above; one double2 of Q per lane and k-step gives both B operands
// ((q.x, q.y) below ph, (-q.y, q.x) above).  A wave owns 16 rows x 64 columns of one frame (four column tiles x {re, im} accumulators), a
// workgroup four such strips; the epilogue is that of k_dft_inv2 (re and im of an element sit in the same lane and register index).
__global__ __launch_bounds__(256) void k_dft_inv2_mfma(const double2 *__restrict__ Q, const double2 *__restrict__ Gy_all, size_t gy_stride,
                                                       double2 *__restrict__ field, float *__restrict__ amp, const double2 *__restrict__ cref_all,
                                                       const float *__restrict__ amp_ref_all, size_t ref_stride, float *__restrict__ prod,
                                                       float *__restrict__ wrapped, int h, int w, int ph)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = (blockIdx.y * 4 + wid) * 16;
    const size_t b = blockIdx.z;
    if (y0 >= h) return;
    const int r = lane & 15, kk = lane >> 4;
    const double *G = (const double *)(Gy_all + b * gy_stride);
    const double2 *Qb = Q + b * (size_t)ph * w;
    const int ya = min(y0 + r, h - 1);
    v4f64 cre[4], cim[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { cre[t] = (v4f64){0.0, 0.0, 0.0, 0.0}; cim[t] = (v4f64){0.0, 0.0, 0.0, 0.0}; }
    const int K = 2 * ph;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + kk;
        const bool in = k < K, hi = k >= ph;
        const int kq = in ? (hi ? k - ph : k) : 0;
        const double a = in ? G[((size_t)ya * ph + kq) * 2 + (hi ? 1 : 0)] : 0.0;
        double2 q[4];
#pragma unroll
        for (int t = 0; t < 4; t++) q[t] = Qb[(size_t)kq * w + min(x0 + 16 * t + r, w - 1)];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const double bre = in ? (hi ? -q[t].y : q[t].x) : 0.0, bim = in ? (hi ? q[t].x : q[t].y) : 0.0;
            cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bre, cre[t], 0, 0, 0);
            cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bim, cim[t], 0, 0, 0);
        }
    }
    const double2 *cref = cref_all ? cref_all + b * ref_stride : nullptr;
    const float *amp_ref = amp_ref_all ? amp_ref_all + b * ref_stride : nullptr;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int x = x0 + 16 * t + r;
        if (x >= w) continue;
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const int y = y0 + kk + 4 * qd;
            if (y >= h) continue;
            const double ar = cre[t][qd], ai = cim[t][qd];
            const size_t p = (size_t)y * w + x, i = b * (size_t)h * w + p;
            if (field) field[i] = make_double2(ar, ai);
            const float am = (float)sqrt(fma(ar, ar, ai * ai));
            amp[i] = am;
            if (cref) {
                const double2 c = cref[p];
                const double rr = ar * c.x + ai * c.y;
                const double ri = ai * c.x - ar * c.y;
                wrapped[i] = (float)atan2(ri, rr);
                prod[i] = __fmul_rn(amp_ref[p], am);
            }
        }
    }
}

void launch_dft_inverse(const double2 *patch, int patch_stride, const double2 *Gx, const double2 *Gy, size_t tab_stride_x, size_t tab_stride_y,
                        double2 *tmpQ, double2 *field, float *amp, const double2 *cref, const float *amp_ref, size_t ref_stride, float *prod,
                        float *wrapped, int B, int h, int w, int ph, int pw, hipStream_t st)
{
    hipLaunchKernelGGL(k_dft_inv1, dim3((w + 255) / 256, ph, B), dim3(256), 0, st, patch, patch_stride, Gx, tab_stride_x, tmpQ, w, ph, pw);
    hipLaunchKernelGGL(k_dft_inv2_mfma, dim3((w + 63) / 64, (h + 63) / 64, B), dim3(256), 0, st, (const double2 *)tmpQ, Gy, tab_stride_y, field, amp, cref,
                       amp_ref, ref_stride, prod, wrapped, h, w, ph);
}

// ---- full spectrum magnitude (reference-frame carrier search, shape_ftp.py:867-872), float64 as np.abs(fft2(float64)) ----
// The frame is real, so F(-f) = conj F(f): only the columns fx = 0 .. Wf/2 are computed (stage 1 = k_dft_fwd1_mfma with the half table
// as its "patch", stage 2 below) and every magnitude is written to its own and to its mirror position of the fftshift-ed plane.
// Stage 2: mag[fy][fx] = | sum_y Eyf[fy][y] * T[y][fx] |, the complex product as the real float64 GEMM of k_dft_inv2_mfma (K = 2h); a wave
// owns 16 frequency rows x 64 columns.
__global__ __launch_bounds__(256) void k_full2_mfma(const double2 *__restrict__ T, const double2 *__restrict__ Eyf, double *__restrict__ mag_all,
                                                    int h, int Hf, int Wf, int Wh)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = (blockIdx.y * 4 + wid) * 16;
    const size_t b = blockIdx.z;
    if (y0 >= Hf) return;
    const int r = lane & 15, kk = lane >> 4;
    const double *G = (const double *)Eyf;
    const double2 *Tb = T + b * (size_t)h * Wh;
    const int ya = min(y0 + r, Hf - 1);
    v4f64 cre[4], cim[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { cre[t] = (v4f64){0.0, 0.0, 0.0, 0.0}; cim[t] = (v4f64){0.0, 0.0, 0.0, 0.0}; }
    const int K = 2 * h;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + kk;
        const bool in = k < K, hi = k >= h;
        const int kq = in ? (hi ? k - h : k) : 0;
        const double a = in ? G[((size_t)ya * h + kq) * 2 + (hi ? 1 : 0)] : 0.0;
        double2 q[4];
#pragma unroll
        for (int t = 0; t < 4; t++) q[t] = Tb[(size_t)kq * Wh + min(x0 + 16 * t + r, Wh - 1)];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const double bre = in ? (hi ? -q[t].y : q[t].x) : 0.0, bim = in ? (hi ? q[t].x : q[t].y) : 0.0;
            cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bre, cre[t], 0, 0, 0);
            cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bim, cim[t], 0, 0, 0);
        }
    }
    double *mag = mag_all + b * (size_t)Hf * Wf;
    const int cy = Hf / 2, cx = Wf / 2;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int fx = x0 + 16 * t + r;
        if (fx >= Wh) continue;
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const int fy = y0 + kk + 4 * qd;
            if (fy >= Hf) continue;
            const double ar = cre[t][qd], ai = cim[t][qd];
            const double m = sqrt(fma(ar, ar, ai * ai));
            mag[(size_t)((fy + cy) % Hf) * Wf + (fx + cx) % Wf] = m;
            const int mx = (Wf - fx) % Wf;                       // mirror column: outside the computed half unless fx is 0 or Wf/2
            if (mx >= Wh) mag[(size_t)(((Hf - fy) % Hf + cy) % Hf) * Wf + (mx + cx) % Wf] = m;
        }
    }
}

// Ex_half: [w][Wh] (Wh = Wf/2 + 1), Ey_full: [Hf][h]; tmp: [B*h][Wh] double2
void launch_dft_full_mag(const float *iw, const float *mu, const double2 *Ex_half, const double2 *Ey_full, double2 *tmp,
                         double *mag, int B, int h, int w, int Hf, int Wf, hipStream_t st)
{
    const int Wh = Wf / 2 + 1, rows = B * h;
    hipLaunchKernelGGL(k_dft_fwd1_mfma, dim3(((rows + 15) / 16 + 3) / 4), dim3(256), 0, st, iw, mu, Ex_half, (size_t)0, tmp, h, w, Wh, rows);
    hipLaunchKernelGGL(k_full2_mfma, dim3((Wh + 63) / 64, (Hf + 63) / 64, B), dim3(256), 0, st, (const double2 *)tmp, Ey_full, mag, h, Hf, Wf, Wh);
}

// top-N magnitudes outside the DC box (find_top_peaks, shape_ftp.py:420-441), descending: out[b][3*i] = x, y, value.  One workgroup
// per frame.  Magnitudes are non-negative doubles: their bit patterns order like the values.  ONE pass over the plane: every thread keeps
// the TP_L largest of its strided elements (value bits, index) in registers; the N largest of the frame are then among the 1024 * TP_L
// survivors unless one thread held more than TP_L of them -- detected (the thread's smallest kept value is then >= the N-th result) and
// handled by the exact N-pass fallback.  Equal values: smaller linear index first.
constexpr int TP_L = 4;
__device__ inline bool tp_before(unsigned long long va, unsigned int ia, unsigned long long vb, unsigned int ib) { return va > vb || (va == vb && ia < ib); }

__global__ __launch_bounds__(1024) void k_top_peaks(const double *__restrict__ mag_all, int Hf, int Wf, int dc, int npeaks, double *__restrict__ out_all)
{
    __shared__ unsigned long long scratch[16];
    __shared__ unsigned int chosen[64];
    __shared__ unsigned long long s_val[1024 * TP_L];
    __shared__ unsigned int s_idx[1024 * TP_L];
    __shared__ int s_overflow;
    const size_t n = (size_t)Hf * Wf;
    const double *mag = mag_all + blockIdx.x * n;
    double *out = out_all + blockIdx.x * (size_t)(3 * 64);
    const int tid = threadIdx.x;
    int cy = Hf / 2, cx = Wf / 2;
    int y0 = max(0, cy - dc), y1 = min(Hf, cy + dc), x0 = max(0, cx - dc), x1 = min(Wf, cx + dc);
    if (npeaks > 64) npeaks = 64;
    // ---- per-thread top TP_L (sorted, best first); `dropped` = the best element this thread had to discard
    unsigned long long tv[TP_L], dropped_v = 0;
    unsigned int ti[TP_L];
    bool dropped = false;
#pragma unroll
    for (int k = 0; k < TP_L; k++) { tv[k] = 0ull; ti[k] = 0xffffffffu; }
    for (size_t i = tid; i < n; i += 1024) {
        const int y = (int)(i / Wf), x = (int)(i % Wf);
        double v = mag[i];
        if (y >= y0 && y < y1 && x >= x0 && x < x1) v = 0.0;
        unsigned long long cv = (unsigned long long)__double_as_longlong(v);
        unsigned int ci = (unsigned int)i;
        if (!tp_before(cv, ci, tv[TP_L - 1], ti[TP_L - 1])) {       // not better than the worst kept: discard
            if (!dropped || cv > dropped_v) dropped_v = cv;
            dropped = true;
            continue;
        }
        if (ti[TP_L - 1] != 0xffffffffu) { if (!dropped || tv[TP_L - 1] > dropped_v) dropped_v = tv[TP_L - 1]; dropped = true; }
#pragma unroll
        for (int k = 0; k < TP_L; k++) {                            // sorted insertion
            if (tp_before(cv, ci, tv[k], ti[k])) { const unsigned long long xv = tv[k]; const unsigned int xi = ti[k]; tv[k] = cv; ti[k] = ci; cv = xv; ci = xi; }
        }
    }
#pragma unroll
    for (int k = 0; k < TP_L; k++) { s_val[tid * TP_L + k] = tv[k]; s_idx[tid * TP_L + k] = ti[k]; }
    if (tid == 0) s_overflow = 0;
    __syncthreads();
    // ---- N rounds of block-wide arg-max over the survivors
    unsigned long long last_v = 0;
    for (int k = 0; k < npeaks; k++) {
        unsigned long long bv = 0;
        unsigned int bi = 0xffffffffu;
        for (int j = tid; j < 1024 * TP_L; j += 1024) {
            const unsigned long long v = s_val[j];
            const unsigned int ix = s_idx[j];
            if (ix != 0xffffffffu && tp_before(v, ix, bv, bi)) { bv = v; bi = ix; }
        }
        const unsigned long long mv = block_max_u64(bv, scratch);
        __syncthreads();
        const unsigned long long mi = block_min_u64(bv == mv && bi != 0xffffffffu ? (unsigned long long)bi : ~0ull, scratch);
        if (tid == 0) {
            chosen[k] = (unsigned int)mi;
            out[3 * k] = (double)((unsigned int)mi % Wf);
            out[3 * k + 1] = (double)((unsigned int)mi / Wf);
            out[3 * k + 2] = __longlong_as_double((long long)mv);
        }
        for (int j = tid; j < 1024 * TP_L; j += 1024)
            if (s_idx[j] == (unsigned int)mi) s_idx[j] = 0xffffffffu;
        last_v = mv;
        __syncthreads();
    }
    // a discarded element that would have made the list: redo exactly (never seen on spectra: the peaks are few and far apart)
    if (dropped && dropped_v >= last_v && last_v > 0) s_overflow = 1;
    __syncthreads();
    if (!s_overflow) return;
    for (int k = 0; k < npeaks; k++) {
        unsigned long long best = 0;
        for (size_t i = tid; i < n; i += 1024) {
            int y = (int)(i / Wf), x = (int)(i % Wf);
            double v = mag[i];
            if (y >= y0 && y < y1 && x >= x0 && x < x1) v = 0.0;
            bool skip = false;
            for (int j = 0; j < k; j++) skip |= (chosen[j] == (unsigned int)i);
            if (skip) continue;
            const unsigned long long key = (unsigned long long)__double_as_longlong(v);
            if (key > best) best = key;
        }
        best = block_max_u64(best, scratch);
        __syncthreads();
        unsigned long long idx = ~0ull;
        for (size_t i = tid; i < n; i += 1024) {
            int y = (int)(i / Wf), x = (int)(i % Wf);
            double v = mag[i];
            if (y >= y0 && y < y1 && x >= x0 && x < x1) v = 0.0;
            bool skip = false;
            for (int j = 0; j < k; j++) skip |= (chosen[j] == (unsigned int)i);
            if (!skip && (unsigned long long)__double_as_longlong(v) == best && i < idx) idx = i;
        }
        idx = block_min_u64(idx, scratch);
        if (tid == 0) {
            chosen[k] = (unsigned int)idx;
            out[3 * k] = (double)(idx % Wf);
            out[3 * k + 1] = (double)(idx / Wf);
            out[3 * k + 2] = __longlong_as_double((long long)best);
        }
        __syncthreads();
    }
}

void launch_top_peaks(const double *mag, int B, int Hf, int Wf, int dc, int npeaks, double *out_xyv, hipStream_t st)
{
    hipLaunchKernelGGL(k_top_peaks, dim3(B), dim3(1024), 0, st, mag, Hf, Wf, dc, npeaks, out_xyv);
}

}  // namespace vf
