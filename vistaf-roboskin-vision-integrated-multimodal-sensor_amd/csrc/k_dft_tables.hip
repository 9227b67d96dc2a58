// Carrier choice and pruned-DFT twiddle tables on the device, per frame (shape_ftp.py:444-483, :930-961).
//
// One code path serves both modes of the library: the session mode (ONE reference frame, tables shared by every
// deformed frame of the session) and the uncached-pair mode (every sample brings its own reference frame and so
// its own carrier: Code/height_to_force.py:384 calls shape_ftp.main per image).
//
//   k_carrier_choose   choose_carrier_peak (:444-463) among the top-N spectrum peaks, refine_peak_parabolic_log
//                      (:473-483) in float64 on the float64 magnitudes (np.abs of a complex128 spectrum), patch
//                      geometry (:930-948), carrier k and period (:907-913)
//   k_build_tables     Ex/Ey (forward, reflect padding of cv2.copyMakeBorder(BORDER_REFLECT) folded in) and
//                      Gx/Gy (inverse: re-centred patch, crop, sub-bin ramp :955-960 and 1/(Hf*Wf) folded in)
// Angles are reduced exactly in integers before sincospi, so a twiddle is good to ~2e-16.
#include "kernels.hpp"

namespace vf {

__global__ void k_carrier_choose(const double *__restrict__ peaks_all, int npk, const double *__restrict__ mag_all, int Hf, int Wf, int bw,
                                 double max_dy_frac, CarrierGeom *__restrict__ geom, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *pk = peaks_all + (size_t)b * 192;
    const double *mag = mag_all + (size_t)b * Hf * Wf;
    const int cys = Hf / 2, cxs = Wf / 2;
    CarrierGeom g;
    // candidate filters: each applies only if it leaves something (shape_ftp.py:447-459); the list is in descending magnitude, so
    // max(cand, key=mag) is its first surviving entry
    unsigned long long cand = npk >= 64 ? ~0ull : ((1ull << npk) - 1ull), f = 0;
    for (int i = 0; i < npk; i++) if (pk[3 * i] > (double)cxs) f |= 1ull << i;
    if (f & cand) cand &= f;
    const int max_dy = (int)(max_dy_frac * Hf);
    f = 0;
    for (int i = 0; i < npk; i++) if (abs((int)pk[3 * i + 1] - cys) <= max_dy) f |= 1ull << i;
    if (f & cand) cand &= f;
    int best = __ffsll((long long)cand) - 1;
    for (int i = 0; i < npk; i++) if (((cand >> i) & 1ull) && pk[3 * i + 2] > pk[3 * best + 2]) best = i;
    const int px = (int)pk[3 * best], py = (int)pk[3 * best + 1];
    g.ok = pk[3 * best + 2] > 0.0 ? 1 : 0;
    double pxf = px, pyf = py;
    if (px > 0 && px < Wf - 1 && py > 0 && py < Hf - 1) {
        auto lg = [&](int yy, int xx) { return log(mag[(size_t)yy * Wf + xx] + 1e-12); };
        auto par = [](double fm1, double f0, double fp1) -> double {
            const double den = (fm1 - 2.0 * f0) + fp1;
            if (fabs(den) < 1e-12) return 0.0;
            return 0.5 * (fm1 - fp1) / den;
        };
        const double c0 = lg(py, px);
        pxf = (double)px + par(lg(py, px - 1), c0, lg(py, px + 1));
        pyf = (double)py + par(lg(py - 1, px), c0, lg(py + 1, px));
    }
    g.px_raw = px; g.py_raw = py;
    g.peak_x = pxf; g.peak_y = pyf;
    g.kx = pxf - cxs; g.ky = pyf - cys;
    g.px_i = (int)rint(pxf); g.py_i = (int)rint(pyf);           // np.round: half to even
    const int x0 = max(0, g.px_i - bw), x1 = min(Wf, g.px_i + bw + 1), y0 = max(0, g.py_i - bw), y1 = min(Hf, g.py_i + bw + 1);
    g.x0 = x0; g.y0 = y0; g.ph = y1 - y0; g.pw = x1 - x0;
    if (g.ph < 1 || g.pw < 1) g.ok = 0;
    double dpx = pxf - g.px_i, dpy = pyf - g.py_i;
    if (!(fabs(dpx) > 1e-6 || fabs(dpy) > 1e-6)) { dpx = 0.0; dpy = 0.0; }
    g.dpx = dpx; g.dpy = dpy;
    g.keep_carrier = 0;
    g.period = fabs(g.kx) > 1e-9 ? (double)Wf / fabs(g.kx) : 0.0;
    if (!isfinite(g.peak_x) || !isfinite(g.peak_y)) g.ok = 0;
    geom[b] = g;
}

void launch_carrier_choose(const double *peaks, int npk, const double *mag, int Hf, int Wf, int bw, double max_dy_frac, CarrierGeom *geom, int B,
                           hipStream_t st)
{
    hipLaunchKernelGGL(k_carrier_choose, dim3((B + 63) / 64), dim3(64), 0, st, peaks, npk, mag, Hf, Wf, bw, max_dy_frac, geom, B);
}

// status2 (may be null): the status words of the pairs' deformed frames when both sets were preprocessed together: folded into status
__global__ void k_pair_status(const CarrierGeom *__restrict__ geom, int pmax, int32_t *__restrict__ status, const int32_t *__restrict__ status2, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const CarrierGeom g = geom[b];
    if (status2 && status[b] == 0 && status2[b] != 0) status[b] = status2[b];
    if (!g.ok || g.ph != pmax || g.pw != pmax || !(g.period > 1e-12)) status[b] = 3;
}
void launch_pair_status(const CarrierGeom *geom, int pmax, int32_t *status, const int32_t *status2, int B, hipStream_t st)
{
    hipLaunchKernelGGL(k_pair_status, dim3((B + 63) / 64), dim3(64), 0, st, geom, pmax, status, status2, B);
}

// exp(sign * 2*pi*i * m / N) for an exactly reduced 0 <= m < N
__device__ inline double2 unit_root(long long m, int N, double sign)
{
    double s, c;
    sincospi(2.0 * (double)m / (double)N, &s, &c);
    return make_double2(c, sign * s);
}
__device__ inline long long posmod(long long a, long long n) { long long r = a % n; return r < 0 ? r + n : r; }

// Sum over the padded positions X in [0, N) that cv's BORDER_REFLECT maps onto source position s (0 <= s < n): X - pad is congruent to
// s or to -s-1 modulo 2n.
template <class F>
__device__ inline void foreach_reflection(int s, int n, int pad, int N, F body)
{
    const int per = 2 * n;
    for (int base = 0; base < 2; base++) {
        const int r = base == 0 ? s : per - s - 1;                 // residue of X - pad modulo 2n
        int X = pad + r - ((pad + r) / per) * per;                 // smallest X >= 0 with X - pad congruent to r
        for (; X < N; X += per) body(X);
    }
}

// grid: (ceil(max(w,h) * pmax / 256), 4 tables, B); table strides in elements
__global__ __launch_bounds__(256) void k_build_tables(const CarrierGeom *__restrict__ geom, int geom_stride, double2 *__restrict__ Ex_all,
                                                      double2 *__restrict__ Ey_all, double2 *__restrict__ Gx_all, double2 *__restrict__ Gy_all,
                                                      size_t sx, size_t sy, int h, int w, int pad, int Hf, int Wf, int pmax)
{
    const size_t b = blockIdx.z;
    const CarrierGeom g = geom[b * geom_stride];
    const int ph = min(g.ph, pmax), pw = min(g.pw, pmax);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int cxs = Wf / 2, cys = Hf / 2;
    switch (blockIdx.y) {
    case 0: {   // Ex[xs][c] = sum_X exp(-2 pi i f_c X / Wf), f_c = x0 + c - cxs
        if (t >= w * pw) return;
        const int xs = t / pw, c = t - xs * pw;
        const long long f = posmod((long long)g.x0 + c - cxs, Wf);
        double er = 0.0, ei = 0.0;
        foreach_reflection(xs, w, pad, Wf, [&](int X) { const double2 u = unit_root((f * X) % Wf, Wf, -1.0); er += u.x; ei += u.y; });
        Ex_all[b * sx + t] = make_double2(er, ei);
        break;
    }
    case 1: {   // Ey[a][ys]
        if (t >= ph * h) return;
        const int a = t / h, ys = t - a * h;
        const long long f = posmod((long long)g.y0 + a - cys, Hf);
        double er = 0.0, ei = 0.0;
        foreach_reflection(ys, h, pad, Hf, [&](int Y) { const double2 u = unit_root((f * Y) % Hf, Hf, -1.0); er += u.x; ei += u.y; });
        Ey_all[b * sy + t] = make_double2(er, ei);
        break;
    }
    case 2: {   // Gx[c][x] = exp(+2 pi i ((c - pw/2) - dpx) (x + pad) / Wf): integer part reduced exactly, sub-bin part |.| <= pi
        if (t >= pw * w) return;
        const int c = t / w, x = t - c * w;
        const long long X = x + pad;
        const long long m = posmod((long long)(g.keep_carrier ? g.x0 + c - cxs : c - pw / 2) * X, Wf);
        double s, co;
        sincospi(2.0 * (double)m / (double)Wf - 2.0 * g.dpx * ((double)X / (double)Wf), &s, &co);
        Gx_all[b * sx + t] = make_double2(co, s);
        break;
    }
    default: {  // Gy[y][a], with the 1 / (Hf * Wf) of ifft2
        if (t >= h * ph) return;
        const int y = t / ph, a = t - y * ph;
        const long long Y = y + pad;
        const long long m = posmod((long long)(g.keep_carrier ? g.y0 + a - cys : a - ph / 2) * Y, Hf);
        double s, co;
        sincospi(2.0 * (double)m / (double)Hf - 2.0 * g.dpy * ((double)Y / (double)Hf), &s, &co);
        const double scale = 1.0 / ((double)Hf * (double)Wf);
        Gy_all[b * sy + t] = make_double2(co * scale, s * scale);
        break;
    }
    }
}

void launch_build_tables(const CarrierGeom *geom, int geom_stride, double2 *Ex, double2 *Ey, double2 *Gx, double2 *Gy, size_t stride_x,
                         size_t stride_y, int B, int h, int w, int pad, int Hf, int Wf, int pmax, hipStream_t st)
{
    const int n = (h > w ? h : w) * pmax;
    hipLaunchKernelGGL(k_build_tables, dim3((n + 255) / 256, 4, B), dim3(256), 0, st, geom, geom_stride, Ex, Ey, Gx, Gy, stride_x, stride_y, h, w,
                       pad, Hf, Wf, pmax);
}

// Full-spectrum tables (carrier search): Exf[xs][f] (w x Wh, Wh = Wf/2 + 1: the spectrum of a real frame is Hermitian), Eyf[f][ys] (Hf x h),
// frame independent.
__global__ __launch_bounds__(256) void k_build_full_tables(double2 *__restrict__ Exf, double2 *__restrict__ Eyf, int h, int w, int pad, int Hf, int Wf)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Wh = Wf / 2 + 1;
    if (blockIdx.y == 0) {
        if (t >= (size_t)w * Wh) return;
        const int xs = (int)(t / Wh), f = (int)(t - (size_t)xs * Wh);
        double er = 0.0, ei = 0.0;
        foreach_reflection(xs, w, pad, Wf, [&](int X) { const double2 u = unit_root(((long long)f * X) % Wf, Wf, -1.0); er += u.x; ei += u.y; });
        Exf[t] = make_double2(er, ei);
    } else {
        if (t >= (size_t)Hf * h) return;
        const int f = (int)(t / h), ys = (int)(t - (size_t)f * h);
        double er = 0.0, ei = 0.0;
        foreach_reflection(ys, h, pad, Hf, [&](int Y) { const double2 u = unit_root(((long long)f * Y) % Hf, Hf, -1.0); er += u.x; ei += u.y; });
        Eyf[t] = make_double2(er, ei);
    }
}

void launch_build_full_tables(double2 *Exf, double2 *Eyf, int h, int w, int pad, int Hf, int Wf, hipStream_t st)
{
    const size_t n = (size_t)(w > h ? w : h) * (size_t)(Wf > Hf ? Wf : Hf);
    hipLaunchKernelGGL(k_build_full_tables, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, st, Exf, Eyf, h, w, pad, Hf, Wf);
}

}  // namespace vf
