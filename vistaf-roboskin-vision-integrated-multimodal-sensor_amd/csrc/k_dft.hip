// Side-band demodulation as a pruned DFT (shape_ftp.py:857-997).
//
// The reference reflect-pads the n x n crop to N x N, takes fft2, keeps a (2*bw+1)^2 patch around the
// (locked) carrier, windows it, moves it to DC, takes ifft2, applies a sub-bin ramp and crops.  Only
// ph*pw (<= 21x21) spectral bins are ever used, so both transforms are computed directly:
//   forward  patch = win . ( Ey[ph x h] . (iw - mu)[h x w] . Ex[w x pw] )       (reflect pad folded into Ex/Ey)
//   inverse  field = Gy[h x ph] . patch[ph x pw] . Gx[pw x w]                   (crop, ramp, 1/N^2 folded in)
// Twiddles are built on the host in double precision; accumulation here is double.
#include "kernels.hpp"

namespace vf {

constexpr int DFT_RB = 12;   // row groups per workgroup in stage 1 (12 * 21 = 252 threads)
constexpr int DFT_RR = 4;    // rows per thread in stage 1: every twiddle loaded feeds DFT_RR rows

// stage 1: T[b, y, c] = sum_x (iw[b,y,x] - mu[b]) * Ex[x, c].  Thread = (row group, c); the rows of the block are staged in LDS.
__global__ __launch_bounds__(256) void k_dft_fwd1(const float *__restrict__ iw, const float *__restrict__ mu,
                                                  const float2 *__restrict__ Ex, double2 *__restrict__ T, int h, int w, int pw, int rb)
{
    extern __shared__ float rows[];   // rb * DFT_RR * w
    size_t b = blockIdx.y;
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
        const float2 e = Ex[(size_t)x * pw + c];
#pragma unroll
        for (int k = 0; k < DFT_RR; k++) {
            const double v = r[k * w + x];
            ar[k] = fma(v, (double)e.x, ar[k]);
            ai[k] = fma(v, (double)e.y, ai[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < DFT_RR; k++)
        if (rg * DFT_RR + k < nr) T[(b * (size_t)h + y0 + rg * DFT_RR + k) * pw + c] = make_double2(ar[k], ai[k]);
}

// stage 2: patch[b, a, c] = win[a,c] * sum_y Ey[a, y] * T[b, y, c]
__global__ void k_dft_fwd2(const double2 *__restrict__ T, const float2 *__restrict__ Ey, const float *__restrict__ win,
                           float2 *__restrict__ patch, int h, int ph, int pw)
{
    size_t b = blockIdx.x;
    int t = threadIdx.x;
    if (t >= ph * pw) return;
    int a = t / pw, c = t % pw;
    const double2 *Tb = T + b * (size_t)h * pw;
    double ar = 0.0, ai = 0.0;
    // one workgroup per frame: the loads of eight rows are issued together (the loop was one memory round trip per row)
    int y = 0;
    for (; y + 8 <= h; y += 8) {
        float2 e[8];
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { e[k] = Ey[(size_t)a * h + y + k]; v[k] = Tb[(size_t)(y + k) * pw + c]; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ar += (double)e[k].x * v[k].x - (double)e[k].y * v[k].y;
            ai += (double)e[k].x * v[k].y + (double)e[k].y * v[k].x;
        }
    }
    for (; y < h; y++) {
        float2 e = Ey[(size_t)a * h + y];
        double2 v = Tb[(size_t)y * pw + c];
        ar += (double)e.x * v.x - (double)e.y * v.y;
        ai += (double)e.x * v.y + (double)e.y * v.x;
    }
    float wv = win[t];
    // complex64 spectrum value times float32 window, as upstream (patch *= win)
    patch[b * (size_t)ph * pw + t] = make_float2(__fmul_rn((float)ar, wv), __fmul_rn((float)ai, wv));
}

void launch_dft_forward(const float *iw, const float *mu, const float2 *Ex, const float2 *Ey, const float *win,
                        float2 *tmpT, float2 *patch, int B, int h, int w, int ph, int pw, hipStream_t st)
{
    int rb = 256 / pw;
    if (rb > DFT_RB) rb = DFT_RB;
    const int rb_lds = (int)((60 * 1024) / ((size_t)DFT_RR * w * sizeof(float)));     // staged rows must fit the default dynamic LDS limit
    if (rb > rb_lds) rb = rb_lds;
    if (rb < 1) rb = 1;
    dim3 g1((h + rb * DFT_RR - 1) / (rb * DFT_RR), B);
    hipLaunchKernelGGL(k_dft_fwd1, g1, dim3(256), (size_t)rb * DFT_RR * w * sizeof(float), st, iw, mu, Ex, (double2 *)tmpT, h, w, pw, rb);
    hipLaunchKernelGGL(k_dft_fwd2, dim3(B), dim3(((ph * pw + 63) / 64) * 64), 0, st, (const double2 *)tmpT, Ey, win, patch, h, ph, pw);
}

// stage 3: Q[b, a, x] = sum_c patch[b, a, c] * Gx[c, x]
__global__ void k_dft_inv1(const float2 *__restrict__ patch, const float2 *__restrict__ Gx, double2 *__restrict__ Q, int w, int ph, int pw)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int a = blockIdx.y;
    size_t b = blockIdx.z;
    if (x >= w) return;
    const float2 *pr = patch + (b * (size_t)ph + a) * pw;
    double ar = 0.0, ai = 0.0;
    for (int c = 0; c < pw; c++) {
        float2 p = pr[c], g = Gx[(size_t)c * w + x];
        ar += (double)p.x * g.x - (double)p.y * g.y;
        ai += (double)p.x * g.y + (double)p.y * g.x;
    }
    Q[(b * (size_t)ph + a) * w + x] = make_double2(ar, ai);
}

// stage 4: field[b, y, x] = sum_a Gy[y, a] * Q[b, a, x];  amp = |field|.  A thread produces DFT_RY rows of one column: every Q word it
// loads feeds DFT_RY outputs (one output per thread re-read the frame's Q once per row through the L2: the kernel ran at L2 bandwidth).
// The Gy rows of the block sit in LDS; the sums run over a in the same order as before: same bits.
constexpr int DFT_RY = 8;
__global__ __launch_bounds__(256) void k_dft_inv2(const double2 *__restrict__ Q, const float2 *__restrict__ Gy, float2 *__restrict__ field,
                                                  float *__restrict__ amp, int h, int w, int ph)
{
    extern __shared__ float2 gy_lds[];                  // [DFT_RY][ph]
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * DFT_RY;
    const size_t b = blockIdx.z;
    for (int i = threadIdx.x; i < DFT_RY * ph; i += blockDim.x) {
        const int r = i / ph, a = i - r * ph;
        gy_lds[i] = y0 + r < h ? Gy[(size_t)(y0 + r) * ph + a] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    if (x >= w) return;
    const double2 *Qb = Q + b * (size_t)ph * w;
    double ar[DFT_RY], ai[DFT_RY];
#pragma unroll
    for (int r = 0; r < DFT_RY; r++) { ar[r] = 0.0; ai[r] = 0.0; }
    for (int a = 0; a < ph; a++) {
        const double2 q = Qb[(size_t)a * w + x];
#pragma unroll
        for (int r = 0; r < DFT_RY; r++) {
            const float2 g = gy_lds[r * ph + a];
            ar[r] += (double)g.x * q.x - (double)g.y * q.y;
            ai[r] += (double)g.x * q.y + (double)g.y * q.x;
        }
    }
#pragma unroll
    for (int r = 0; r < DFT_RY; r++) {
        if (y0 + r >= h) break;
        const size_t i = b * (size_t)h * w + (size_t)(y0 + r) * w + x;
        field[i] = make_float2((float)ar[r], (float)ai[r]);
        amp[i] = (float)sqrt(ar[r] * ar[r] + ai[r] * ai[r]);
    }
}

void launch_dft_inverse(const float2 *patch, const float2 *Gx, const float2 *Gy, float2 *tmpQ, float2 *field, float *amp,
                        int B, int h, int w, int ph, int pw, hipStream_t st)
{
    hipLaunchKernelGGL(k_dft_inv1, dim3((w + 255) / 256, ph, B), dim3(256), 0, st, patch, Gx, (double2 *)tmpQ, w, ph, pw);
    hipLaunchKernelGGL(k_dft_inv2, dim3((w + 255) / 256, (h + DFT_RY - 1) / DFT_RY, B), dim3(256), (size_t)DFT_RY * ph * sizeof(float2), st,
                       (const double2 *)tmpQ, Gy, field, amp, h, w, ph);
}

// ---- full spectrum magnitude of ONE frame (reference-frame carrier search, shape_ftp.py:867-872) ----
__global__ void k_full1(const float *__restrict__ iw, const float *__restrict__ mu, const float2 *__restrict__ Ex, double2 *__restrict__ T,
                        int h, int w, int Wf)
{
    int fx = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (fx >= Wf) return;
    const float *r = iw + (size_t)y * w;
    float m = mu[0];
    double ar = 0.0, ai = 0.0;
    for (int x = 0; x < w; x++) {
        float2 e = Ex[(size_t)x * Wf + fx];
        double v = __fsub_rn(r[x], m);
        ar = fma(v, (double)e.x, ar);
        ai = fma(v, (double)e.y, ai);
    }
    T[(size_t)y * Wf + fx] = make_double2(ar, ai);
}

// mag in fftshift layout; the DC exclusion box is zeroed as find_top_peaks does (shape_ftp.py:425-430)
__global__ void k_full2(const double2 *__restrict__ T, const float2 *__restrict__ Ey, float *__restrict__ mag, int h, int Hf, int Wf, int dc)
{
    int sx = blockIdx.x * blockDim.x + threadIdx.x;
    int sy = blockIdx.y;
    if (sx >= Wf) return;
    int cy = Hf / 2, cx = Wf / 2;
    int fy = (sy - cy + Hf) % Hf, fx = (sx - cx + Wf) % Wf;
    double ar = 0.0, ai = 0.0;
    for (int y = 0; y < h; y++) {
        float2 e = Ey[(size_t)fy * h + y];
        double2 v = T[(size_t)y * Wf + fx];
        ar += (double)e.x * v.x - (double)e.y * v.y;
        ai += (double)e.x * v.y + (double)e.y * v.x;
    }
    float m = (float)sqrt(ar * ar + ai * ai);
    (void)dc;
    mag[(size_t)sy * Wf + sx] = m;
}

void launch_dft_full_mag(const float *iw, const float *mu, const float2 *Ex_full, const float2 *Ey_full, float2 *tmp,
                         float *mag, int h, int w, int Hf, int Wf, int dc_excl, hipStream_t st)
{
    hipLaunchKernelGGL(k_full1, dim3((Wf + 255) / 256, h), dim3(256), 0, st, iw, mu, Ex_full, (double2 *)tmp, h, w, Wf);
    hipLaunchKernelGGL(k_full2, dim3((Wf + 255) / 256, Hf), dim3(256), 0, st, (const double2 *)tmp, Ey_full, mag, h, Hf, Wf, dc_excl);
}

// top-N magnitudes outside the DC box, descending: out[3*i] = x, y, value
__global__ __launch_bounds__(1024) void k_top_peaks(const float *__restrict__ mag, int Hf, int Wf, int dc, int npeaks, float *__restrict__ out)
{
    __shared__ unsigned long long scratch[16];
    __shared__ unsigned int chosen[64];
    int cy = Hf / 2, cx = Wf / 2;
    int y0 = max(0, cy - dc), y1 = min(Hf, cy + dc), x0 = max(0, cx - dc), x1 = min(Wf, cx + dc);
    size_t n = (size_t)Hf * Wf;
    for (int k = 0; k < npeaks && k < 64; k++) {
        unsigned long long best = 0;
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
            int y = (int)(i / Wf), x = (int)(i % Wf);
            float v = mag[i];
            if (y >= y0 && y < y1 && x >= x0 && x < x1) v = 0.f;
            bool skip = false;
            for (int j = 0; j < k; j++) skip |= (chosen[j] == (unsigned int)i);
            if (skip) continue;
            unsigned long long key = ((unsigned long long)f2key(v) << 32) | (unsigned int)(0xffffffffu - (unsigned int)i);
            if (key > best) best = key;
        }
        best = block_max_u64(best, scratch);
        unsigned int idx = 0xffffffffu - (unsigned int)(best & 0xffffffffu);
        if (threadIdx.x == 0) {
            chosen[k] = idx;
            out[3 * k] = (float)(idx % Wf);
            out[3 * k + 1] = (float)(idx / Wf);
            out[3 * k + 2] = key2f((unsigned int)(best >> 32));
        }
        __syncthreads();
    }
}

void launch_top_peaks(const float *mag, int Hf, int Wf, int dc, int npeaks, float *out_xyv, hipStream_t st)
{
    hipLaunchKernelGGL(k_top_peaks, dim3(1), dim3(1024), 0, st, mag, Hf, Wf, dc, npeaks, out_xyv);
}

// ---- phase difference (shape_ftp.py:742, :1681, :1689) -------------------------------------------
__global__ void k_phase_diff(const float2 *__restrict__ cdef, const float2 *__restrict__ cref, const float *__restrict__ amp_def,
                             const float *__restrict__ amp_ref, float *__restrict__ prod, float *__restrict__ wrapped, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float2 d = cdef[i], r = cref[p];
    double rr = (double)d.x * r.x + (double)d.y * r.y;
    double ri = (double)d.y * r.x - (double)d.x * r.y;
    wrapped[i] = (float)atan2(ri, rr);
    prod[i] = __fmul_rn(amp_ref[p], amp_def[i]);
}

void launch_phase_diff(const float2 *cdef, const float2 *cref, const float *amp_def, const float *amp_ref, float *prod,
                       float *wrapped, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_phase_diff, dim3((P + 255) / 256, B), dim3(256), 0, st, cdef, cref, amp_def, amp_ref, prod, wrapped, P);
}

}  // namespace vf
