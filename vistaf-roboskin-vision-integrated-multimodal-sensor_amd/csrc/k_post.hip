// Post-unwrap stages of shape_ftp.main (:1716-1873) and the force tail (force_sensor.py:93-167):
// contact / background masks, masked smoothing glue, sign flip, frontier taper, composition,
// unitless -> mm curve, blob filter, volume / area / max-depth / force / arg-extremum reductions.
#include "kernels.hpp"

namespace vf {

__device__ inline float nanf32() { return __uint_as_float(0x7fc00000u); }

// ---- contact mask (shape_ftp.py:1719-1732) -----------------------------------------------------
// counts[b] = #{reliable & finite & |res| >= thr3[b,0]}
__global__ void k_contact_count(const float *__restrict__ res, const uint8_t *__restrict__ reliable, const float *__restrict__ thr3,
                                int *__restrict__ counts, int P)
{
    __shared__ int scratch[16];
    size_t b = blockIdx.y;
    float thr = thr3[b * 3];
    if (!finitef(thr)) thr = thr3[b * 3 + 1];
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        float v = fabsf(res[b * (size_t)P + p]);
        c += (reliable[b * (size_t)P + p] && finitef(v) && v >= thr);
    }
    c = block_sum<int>(c, scratch);
    if (threadIdx.x == 0 && c) atomicAdd(&counts[b], c);
}

__global__ void k_contact_mask(const float *__restrict__ res, const uint8_t *__restrict__ reliable, const float *__restrict__ thr3,
                               const int *__restrict__ contact_count, const int *__restrict__ rel_count, float min_frac, float max_frac,
                               uint8_t *__restrict__ contact, float *__restrict__ thr_used, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    float thr = thr3[b * 3];
    if (!finitef(thr)) thr = thr3[b * 3 + 1];
    int rc = rel_count[b];
    double frac = (double)contact_count[b] / (double)(rc > 1 ? rc : 1);
    if (frac < (double)min_frac) { float t2 = thr3[b * 3 + 1]; if (finitef(t2)) thr = t2; }
    else if (frac > (double)max_frac) { float t2 = thr3[b * 3 + 2]; if (finitef(t2)) thr = t2; }
    if (p == 0) thr_used[b] = thr;
    if (p >= P) return;
    float v = fabsf(res[b * (size_t)P + p]);
    contact[b * (size_t)P + p] = (uint8_t)(reliable[b * (size_t)P + p] && finitef(v) && v >= thr);
}

void launch_contact_mask(const float *res, const uint8_t *reliable, const float *thr3, const int *rel_count, int *contact_count,
                         float min_frac, float max_frac, uint8_t *contact, float *thr_used, int B, int P, hipStream_t st)
{
    hipMemsetAsync(contact_count, 0, sizeof(int) * B, st);
    int gx = (P + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(k_contact_count, dim3(gx, B), dim3(256), 0, st, res, reliable, thr3, contact_count, P);
    hipLaunchKernelGGL(k_contact_mask, dim3((P + 255) / 256, B), dim3(256), 0, st, res, reliable, thr3, contact_count, rel_count, min_frac,
                       max_frac, contact, thr_used, P);
}

// ---- background = reliable & ~contact_d, falling back to reliable (shape_ftp.py:1738-1741) --------
__global__ void k_and_not(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)(a[i] && !b[i]);
}
__global__ void k_background_fix(const uint8_t *__restrict__ reliable, const int *__restrict__ rel_count, const int *__restrict__ bg_count,
                                 uint8_t *__restrict__ bg, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    int limit = (int)(0.15 * (double)rel_count[b]);
    if (bg_count[b] < limit) bg[b * (size_t)P + p] = reliable[b * (size_t)P + p];
}
void launch_background(const uint8_t *reliable, const uint8_t *contact_d, const int *rel_count, int *bg_count, uint8_t *background,
                       int B, int P, hipStream_t st)
{
    size_t n = (size_t)B * P;
    hipLaunchKernelGGL(k_and_not, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reliable, contact_d, background, n);
    launch_count_u8(background, bg_count, B, P, st);
    hipLaunchKernelGGL(k_background_fix, dim3((P + 255) / 256, B), dim3(256), 0, st, reliable, rel_count, bg_count, background, P);
}

// ---- zeroed = detrended - bg_med ; z0 = zeroed on (mask & finite) else 0 ; m = that mask as float ----
// (shape_ftp.py:1750, :1756, :1142-1144)
__global__ void k_sub_scalar_mask(const float *__restrict__ src, const float *__restrict__ scalar, const uint8_t *__restrict__ mask,
                                  float *__restrict__ z0, float *__restrict__ m_out, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float v = __fsub_rn(src[i], scalar[b]);
    bool ok = mask[i] && finitef(v);
    z0[i] = ok ? v : 0.f;
    m_out[i] = ok ? 1.f : 0.f;
}
void launch_sub_scalar_mask(const float *src, const float *scalar, const uint8_t *mask, float *z0, float *m_out, int B, int P,
                            hipStream_t st)
{
    hipLaunchKernelGGL(k_sub_scalar_mask, dim3((P + 255) / 256, B), dim3(256), 0, st, src, scalar, mask, z0, m_out, P);
}

// num / (den + 1e-6)   (shape_ftp.py:1146-1147)
__global__ void k_div_planes(const float *__restrict__ num, const float *__restrict__ den, float *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __fdiv_rn(num[i], __fadd_rn(den[i], 1e-6f));
}
void launch_div_planes(const float *num, const float *den, float *out, int B, int P, hipStream_t st)
{
    size_t n = (size_t)B * P;
    hipLaunchKernelGGL(k_div_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, num, den, out, n);
}

// ---- auto sign flip (shape_ftp.py:1759-1768) ----------------------------------------------------
__global__ void k_core_flip(float *__restrict__ hmap, const float *__restrict__ core_med, int *__restrict__ flipped, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    float m = core_med[b];
    bool flip = finitef(m) && m > 0.f;
    if (p == 0) flipped[b] = flip ? 1 : 0;
    if (p >= P || !flip) return;
    hmap[b * (size_t)P + p] = __fmul_rn(hmap[b * (size_t)P + p], -1.0f);
}
void launch_core_flip(float *hmap, const float *core_med, int *flipped, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_core_flip, dim3((P + 255) / 256, B), dim3(256), 0, st, hmap, core_med, flipped, P);
}

// ---- frontier taper inside reliable + composition (shape_ftp.py:1770-1818, :1287-1318) -----------
// z0 = height_final with NaN -> 0: 0 on unreliable ROI / outside ROI, tapered height on reliable.
__global__ void k_frontier_compose(const float *__restrict__ hmap, const uint8_t *__restrict__ reliable, const uint8_t *__restrict__ roi,
                                   const float *__restrict__ dist_in, float band, float *__restrict__ z0, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float v = 0.f;
    if (reliable[i] && roi[p]) {
        const float hgt = hmap[i];        // finite: `reliable` is output_reliable = reliable & isfinite(height) (:1801)
        float de = fmaxf(__fsub_rn(dist_in[i], 1.0f), 0.0f);
        float t = __fdiv_rn(de, fmaxf(1e-6f, band));
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        float wgt = __fmul_rn(__fmul_rn(t, t), __fsub_rn(3.0f, __fmul_rn(2.0f, t)));
        v = __fmul_rn(hgt, wgt);
    }
    z0[i] = v;
}
void launch_frontier_compose(const float *hmap, const uint8_t *reliable, const uint8_t *roi, const float *dist_in, float band,
                             float *hfinal_z0, int B, int P, hipStream_t st)
{
    hipLaunchKernelGGL(k_frontier_compose, dim3((P + 255) / 256, B), dim3(256), 0, st, hmap, reliable, roi, dist_in, band, hfinal_z0, P);
}

// unreliable ROI <- masked blur, outside band <- 0, clamp positives, NaN outside ROI (:1820-1841)
__global__ void k_finalize_unitless(const float *__restrict__ z0, const float *__restrict__ smooth_num, const float *__restrict__ roi_den,
                                    const uint8_t *__restrict__ reliable, const uint8_t *__restrict__ roi, const float *__restrict__ dist_out,
                                    float band, int use_band, float *__restrict__ unitless, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    float v = nanf32();
    if (roi[p]) {
        bool rel = reliable[i] != 0;
        v = z0[i];
        if (!rel) {
            if (smooth_num) v = __fdiv_rn(smooth_num[i], roi_den[p]);
            if (use_band) {
                float de = fmaxf(__fsub_rn(dist_out[i], 1.0f), 0.0f);
                if (de <= band) v = 0.f;
            }
        }
        if (finitef(v)) v = fminf(v, 0.0f);
    }
    unitless[i] = v;
}
void launch_finalize_unitless(const float *hfinal_z0, const float *smooth_num, const float *roi_den, const uint8_t *reliable,
                              const uint8_t *roi, const float *dist_out, float band, int use_band, float *unitless, int B, int P,
                              hipStream_t st)
{
    hipLaunchKernelGGL(k_finalize_unitless, dim3((P + 255) / 256, B), dim3(256), 0, st, hfinal_z0, smooth_num, roi_den, reliable, roi,
                       dist_out, band, use_band, unitless, P);
}

// ---- unitless -> mm (shape_ftp.py:682-705) + blob candidates + per-frame max (:1232-1236) --------
__global__ void k_to_mm(const float *__restrict__ unitless, const uint8_t *__restrict__ roi, Curve curve, int use_neg,
                        float *__restrict__ depth, uint8_t *__restrict__ cand, unsigned int *__restrict__ gmax_bits, int P)
{
    __shared__ unsigned long long scratch[16];
    size_t b = blockIdx.y;
    unsigned int mx = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        size_t i = b * (size_t)P + p;
        float hgt = unitless[i];
        float d = hgt;   // NaN stays NaN
        if (hgt == hgt) {
            double x = use_neg ? -(double)hgt : (double)hgt;
            x = fmax(x, 0.0);
            d = (float)curve_eval(curve, x);
        }
        depth[i] = d;
        bool c = roi[p] && finitef(d) && d > 0.0f;
        cand[i] = (uint8_t)c;
        if (c) { unsigned int u = __float_as_uint(d); if (u > mx) mx = u; }
    }
    unsigned long long m = block_max_u64(mx, scratch);
    if (threadIdx.x == 0 && m) atomicMax(&gmax_bits[b], (unsigned int)m);
}
void launch_to_mm(const float *unitless, const uint8_t *roi, Curve curve, int use_neg, float *depth, uint8_t *cand,
                  unsigned int *gmax_bits, int B, int P, hipStream_t st)
{
    hipMemsetAsync(gmax_bits, 0, sizeof(unsigned int) * B, st);
    int gx = (P + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(k_to_mm, dim3(gx, B), dim3(256), 0, st, unitless, roi, curve, use_neg, depth, cand, gmax_bits, P);
}

// ---- blob filter (shape_ftp.py:1238-1271): per-component peak, keep if peak >= thr ------------------
__global__ void k_blob_peaks(const float *__restrict__ depth, const int32_t *__restrict__ labels, unsigned int *__restrict__ peak_bits, int P)
{
    size_t b = blockIdx.y;
    int lane = threadIdx.x & 63;
    int Pr = ((P + 255) / 256) * 256;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Pr; p += gridDim.x * blockDim.x) {
        int root = p < P ? labels[b * (size_t)P + p] : -1;
        unsigned int v = root >= 0 ? __float_as_uint(depth[b * (size_t)P + p]) : 0u;
        unsigned long long active = __ballot(root >= 0);
        while (active) {
            int leader = __ffsll((long long)active) - 1;
            int r0 = __shfl(root, leader, 64);
            bool same = root == r0;
            unsigned int m = wave_max_u32(same ? v : 0u);
            if (lane == leader) atomicMax(&peak_bits[b * (size_t)P + r0], m);
            active &= ~__ballot(same);
        }
    }
}
__global__ void k_blob_apply(float *__restrict__ depth, const uint8_t *__restrict__ cand, const int32_t *__restrict__ labels,
                             const unsigned int *__restrict__ peak_bits, const unsigned int *__restrict__ gmax_bits, float min_peak_mm,
                             double rel_frac, uint8_t *__restrict__ kept, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    bool k = false;
    if (cand[i]) {
        double gmax = (double)__uint_as_float(gmax_bits[b]);
        double thr = (double)min_peak_mm;
        if (rel_frac >= 0.0) thr = fmax(thr, rel_frac * gmax);
        double peak = (double)__uint_as_float(peak_bits[b * (size_t)P + labels[i]]);
        k = peak >= thr;
        if (!k) depth[i] = 0.f;
    }
    if (kept) kept[i] = (uint8_t)k;
}
void launch_blob_filter(float *depth, const uint8_t *cand, const int32_t *labels, unsigned int *peak_bits,
                        const unsigned int *gmax_bits, float min_peak_mm, double rel_frac, uint8_t *kept, int B, int P,
                        hipStream_t st)
{
    hipMemsetAsync(peak_bits, 0, sizeof(unsigned int) * (size_t)B * P, st);
    int gx = (P + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(k_blob_peaks, dim3(gx, B), dim3(256), 0, st, depth, labels, peak_bits, P);
    hipLaunchKernelGGL(k_blob_apply, dim3((P + 255) / 256, B), dim3(256), 0, st, depth, cand, labels, peak_bits, gmax_bits, min_peak_mm,
                       rel_frac, kept, P);
}

// ---- force tail + arg-extrema, one 1024-thread workgroup per frame --------------------------------
// depth_map_to_volume_cm3 (force_sensor.py:93-123) with roi = roi_frame / isfinite(height)
// (multimodal_sensor.py:388); predict_force_from_volume (:149-167); nanargmax of depth over ROI
// (shape_ftp.py:1945-1959); argmin of the unitless height (phase_to_height.py:1009-1016).
__global__ __launch_bounds__(1024) void k_tail(const float *__restrict__ height_mm, const uint8_t *__restrict__ roi_frame,
                                               const float *__restrict__ unitless, const uint8_t *__restrict__ roi_static, PostParams pp,
                                               double *__restrict__ scalars, int nscal, double *__restrict__ out3, int P)
{
    __shared__ double sd[16];
    __shared__ unsigned long long s64[16];
    size_t b = blockIdx.x;
    const float *H = height_mm + b * (size_t)P;
    const uint8_t *R = roi_frame ? roi_frame + b * (size_t)P : nullptr;
    // ONE pass over the planes, four pixels per thread in flight: the dominant sign (nansum(neg) > nansum(pos); float32 sums upstream,
    // double here) is only known at the end, so the volume / area / maximum are accumulated for both signs and the right set is kept.
    // Per thread the pixels come in the same order as in separate passes: the sums are the same bits.
    const float *U = unitless ? unitless + b * (size_t)P : nullptr;
    const bool want_arg = scalars != nullptr;
    const float eps = (float)pp.depth_eps_mm;
    double sp = 0, sn = 0, volp = 0, voln = 0;
    int cntp = 0, cntn = 0;
    unsigned long long mxp = 0, mxn = 0, am = 0, an = ~0ull;
    constexpr int TU = 4;
    const int T = blockDim.x;
    for (int p0 = threadIdx.x; p0 < P; p0 += TU * T) {
        float v[TU], u[TU];
        uint8_t rf[TU], rs[TU];
#pragma unroll
        for (int k = 0; k < TU; k++) {
            const int p = p0 + k * T;
            const bool inb = p < P;
            v[k] = inb ? H[p] : nanf32();
            rf[k] = (inb && R) ? R[p] : (uint8_t)0;
            rs[k] = (inb && want_arg) ? roi_static[p] : (uint8_t)0;
            u[k] = (inb && U) ? U[p] : nanf32();
        }
#pragma unroll
        for (int k = 0; k < TU; k++) {
            const int p = p0 + k * T;
            if (p >= P) break;
            const float vv = v[k];
            if (vv == vv) { if (vv > 0.f) sp += vv; else sn += -vv; }
            const bool in = R ? rf[k] != 0 : finitef(vv);
            float dp = fmaxf(vv, 0.f), dn = fmaxf(-vv, 0.f);
            if (!in || !finitef(dp)) dp = 0.f;
            if (!in || !finitef(dn)) dn = 0.f;
            if (dp > eps) { volp += dp; cntp++; const unsigned long long key = (unsigned long long)__float_as_uint(dp) << 32; if (key > mxp) mxp = key; }
            if (dn > eps) { voln += dn; cntn++; const unsigned long long key = (unsigned long long)__float_as_uint(dn) << 32; if (key > mxn) mxn = key; }
            if (rs[k] && finitef(vv)) {            // arg-max of depth (mm) over roi & finite: first occurrence of the maximum
                const unsigned long long key = ((unsigned long long)f2key(vv) << 32) | (unsigned int)(0xffffffffu - (unsigned int)p);
                if (key > am) am = key;
            }
            if (rs[k] && finitef(u[k])) {          // arg-min of unitless height over roi & finite: first occurrence of the minimum
                const unsigned long long key = ((unsigned long long)f2key(u[k]) << 32) | (unsigned int)p;
                if (key < an) an = key;
            }
        }
    }
    sp = block_sum<double>(sp, sd);
    sn = block_sum<double>(sn, sd);
    const bool use_neg = (float)sn > (float)sp;
    double vol = block_sum<double>(use_neg ? voln : volp, sd);
    double cntd = block_sum<double>((double)(use_neg ? cntn : cntp), sd);
    unsigned long long mx = block_max_u64(use_neg ? mxn : mxp, s64);
    double period_px = pp.period_px, mm_per_px = pp.mm_per_px;
    if (pp.pair_geom) { period_px = pp.pair_geom[b].period; mm_per_px = period_px > 1e-12 ? pp.grating_pitch_mm / period_px : 0.0; }
    double area_px = mm_per_px * mm_per_px;
    double volume_cm3 = cntd > 0 ? (double)(float)vol * area_px / 1000.0 : 0.0;
    double area_mm2 = cntd * area_px;
    double maxd = cntd > 0 ? (double)__uint_as_float((unsigned int)(mx >> 32)) : 0.0;
    if (out3 && threadIdx.x == 0) { out3[b * 3] = volume_cm3; out3[b * 3 + 1] = area_mm2; out3[b * 3 + 2] = maxd; }
    if (!scalars) return;
    am = block_max_u64(am, s64);
    if (unitless) an = block_min_u64(an, s64);
    if (threadIdx.x == 0) {
        double *S = scalars + b * (size_t)nscal;
        S[0] = volume_cm3; S[1] = area_mm2; S[2] = maxd;
        S[3] = curve_eval(pp.force_curve, volume_cm3);
        S[4] = am ? (double)(0xffffffffu - (unsigned int)(am & 0xffffffffu)) : -1.0;
        S[5] = period_px; S[6] = mm_per_px;
        if (unitless && an != ~0ull) { S[7] = (double)key2f((unsigned int)(an >> 32)); S[8] = (double)(unsigned int)(an & 0xffffffffu); }
        else { S[7] = (double)nanf32(); S[8] = -1.0; }
    }
}
// The same tail for frames where one workgroup per frame leaves the chip idle (eight native crops: 1.1 ms): k_tail_part accumulates a block's
// 4096 pixels exactly as k_tail accumulates a thread's, k_tail_final adds the blocks' partial results up in a fixed order (deterministic;
// the float64 sums are rounded once to float32 as before) and finishes as k_tail does.
constexpr int TP_T = 256, TP_PX = 16, TP_WORDS = 10;      // partial record: sp, sn, volp, voln (f64), cntp, cntn, mxp, mxn, am, an (u64)
__global__ __launch_bounds__(TP_T) void k_tail_part(const float *__restrict__ height_mm, const uint8_t *__restrict__ roi_frame,
                                                    const float *__restrict__ unitless, const uint8_t *__restrict__ roi_static, float eps, int want_arg,
                                                    unsigned long long *__restrict__ part, int P)
{
    __shared__ double sd[16];
    __shared__ unsigned long long s64[16];
    const size_t b = blockIdx.y;
    const float *H = height_mm + b * (size_t)P;
    const uint8_t *R = roi_frame ? roi_frame + b * (size_t)P : nullptr;
    const float *U = unitless ? unitless + b * (size_t)P : nullptr;
    double sp = 0, sn = 0, volp = 0, voln = 0;
    unsigned long long cntp = 0, cntn = 0, mxp = 0, mxn = 0, am = 0, an = ~0ull;
    const int p0 = blockIdx.x * TP_T * TP_PX + threadIdx.x;
#pragma unroll 4
    for (int k = 0; k < TP_PX; k++) {
        const int p = p0 + k * TP_T;
        if (p >= P) break;
        const float vv = H[p];
        const uint8_t rf = R ? R[p] : (uint8_t)0, rs = want_arg ? roi_static[p] : (uint8_t)0;
        const float uu = U ? U[p] : nanf32();
        if (vv == vv) { if (vv > 0.f) sp += vv; else sn += -vv; }
        const bool in = R ? rf != 0 : finitef(vv);
        float dp = fmaxf(vv, 0.f), dn = fmaxf(-vv, 0.f);
        if (!in || !finitef(dp)) dp = 0.f;
        if (!in || !finitef(dn)) dn = 0.f;
        if (dp > eps) { volp += dp; cntp++; const unsigned long long key = (unsigned long long)__float_as_uint(dp) << 32; if (key > mxp) mxp = key; }
        if (dn > eps) { voln += dn; cntn++; const unsigned long long key = (unsigned long long)__float_as_uint(dn) << 32; if (key > mxn) mxn = key; }
        if (rs && finitef(vv)) {
            const unsigned long long key = ((unsigned long long)f2key(vv) << 32) | (unsigned int)(0xffffffffu - (unsigned int)p);
            if (key > am) am = key;
        }
        if (rs && finitef(uu)) {
            const unsigned long long key = ((unsigned long long)f2key(uu) << 32) | (unsigned int)p;
            if (key < an) an = key;
        }
    }
    sp = block_sum<double>(sp, sd); sn = block_sum<double>(sn, sd);
    volp = block_sum<double>(volp, sd); voln = block_sum<double>(voln, sd);
    cntp = block_sum<unsigned long long>(cntp, s64); cntn = block_sum<unsigned long long>(cntn, s64);
    mxp = block_max_u64(mxp, s64); mxn = block_max_u64(mxn, s64);
    am = block_max_u64(am, s64); an = block_min_u64(an, s64);
    if (threadIdx.x == 0) {
        unsigned long long *o = part + (b * gridDim.x + blockIdx.x) * TP_WORDS;
        o[0] = (unsigned long long)__double_as_longlong(sp); o[1] = (unsigned long long)__double_as_longlong(sn);
        o[2] = (unsigned long long)__double_as_longlong(volp); o[3] = (unsigned long long)__double_as_longlong(voln);
        o[4] = cntp; o[5] = cntn; o[6] = mxp; o[7] = mxn; o[8] = am; o[9] = an;
    }
}
__global__ __launch_bounds__(TP_T) void k_tail_final(const unsigned long long *__restrict__ part, int nblk, int has_unitless, PostParams pp,
                                                     double *__restrict__ scalars, int nscal, double *__restrict__ out3)
{
    __shared__ double sd[16];
    __shared__ unsigned long long s64[16];
    const size_t b = blockIdx.x;
    double sp = 0, sn = 0, volp = 0, voln = 0;
    unsigned long long cntp = 0, cntn = 0, mxp = 0, mxn = 0, am = 0, an = ~0ull;
    for (int k = threadIdx.x; k < nblk; k += TP_T) {
        const unsigned long long *o = part + (b * nblk + k) * TP_WORDS;
        sp += __longlong_as_double((long long)o[0]); sn += __longlong_as_double((long long)o[1]);
        volp += __longlong_as_double((long long)o[2]); voln += __longlong_as_double((long long)o[3]);
        cntp += o[4]; cntn += o[5];
        mxp = o[6] > mxp ? o[6] : mxp; mxn = o[7] > mxn ? o[7] : mxn; am = o[8] > am ? o[8] : am; an = o[9] < an ? o[9] : an;
    }
    sp = block_sum<double>(sp, sd);
    sn = block_sum<double>(sn, sd);
    const bool use_neg = (float)sn > (float)sp;
    const double vol = block_sum<double>(use_neg ? voln : volp, sd);
    const double cntd = (double)block_sum<unsigned long long>(use_neg ? cntn : cntp, s64);
    const unsigned long long mx = block_max_u64(use_neg ? mxn : mxp, s64);
    double period_px = pp.period_px, mm_per_px = pp.mm_per_px;
    if (pp.pair_geom) { period_px = pp.pair_geom[b].period; mm_per_px = period_px > 1e-12 ? pp.grating_pitch_mm / period_px : 0.0; }
    const double area_px = mm_per_px * mm_per_px;
    const double volume_cm3 = cntd > 0 ? (double)(float)vol * area_px / 1000.0 : 0.0;
    const double area_mm2 = cntd * area_px;
    const double maxd = cntd > 0 ? (double)__uint_as_float((unsigned int)(mx >> 32)) : 0.0;
    if (out3 && threadIdx.x == 0) { out3[b * 3] = volume_cm3; out3[b * 3 + 1] = area_mm2; out3[b * 3 + 2] = maxd; }
    if (!scalars) return;
    am = block_max_u64(am, s64);
    an = block_min_u64(an, s64);
    if (threadIdx.x == 0) {
        double *S = scalars + b * (size_t)nscal;
        S[0] = volume_cm3; S[1] = area_mm2; S[2] = maxd;
        S[3] = curve_eval(pp.force_curve, volume_cm3);
        S[4] = am ? (double)(0xffffffffu - (unsigned int)(am & 0xffffffffu)) : -1.0;
        S[5] = period_px; S[6] = mm_per_px;
        if (has_unitless && an != ~0ull) { S[7] = (double)key2f((unsigned int)(an >> 32)); S[8] = (double)(unsigned int)(an & 0xffffffffu); }
        else { S[7] = (double)nanf32(); S[8] = -1.0; }
    }
}

// scratch (optional): room for the per-block records of the chain above, `scratch_bytes` long; without it, or for small frames: k_tail
void launch_tail(const float *height_mm, const uint8_t *roi_or_null, const float *unitless_or_null, const uint8_t *roi_static,
                 PostParams pp, double *scalars, int nscal, double *out3_or_null, int B, int P, hipStream_t st, void *scratch, size_t scratch_bytes)
{
    const int nblk = (P + TP_T * TP_PX - 1) / (TP_T * TP_PX);
    if (scratch && P >= 262144 && B <= 192 && (size_t)B * nblk * TP_WORDS * 8 <= scratch_bytes) {
        unsigned long long *part = (unsigned long long *)scratch;
        hipLaunchKernelGGL(k_tail_part, dim3(nblk, B), dim3(TP_T), 0, st, height_mm, roi_or_null, unitless_or_null, roi_static, (float)pp.depth_eps_mm,
                           scalars != nullptr ? 1 : 0, part, P);
        hipLaunchKernelGGL(k_tail_final, dim3(B), dim3(TP_T), 0, st, part, nblk, unitless_or_null != nullptr ? 1 : 0, pp, scalars, nscal, out3_or_null);
        return;
    }
    hipLaunchKernelGGL(k_tail, dim3(B), dim3(1024), 0, st, height_mm, roi_or_null, unitless_or_null, roi_static, pp, scalars, nscal,
                       out3_or_null, P);
}

// scatter small per-frame values into the scalar record
__global__ void k_fill_scalars(double *__restrict__ scalars, int nscal, const int *__restrict__ rel_count, const int *__restrict__ flipped,
                               const float *__restrict__ amp_thr, const float *__restrict__ contact_thr, const float *__restrict__ bg_med,
                               const int *__restrict__ bad_count, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double *S = scalars + (size_t)b * nscal;
    S[9] = rel_count ? (double)rel_count[b] : 0.0;
    S[10] = flipped ? (double)flipped[b] : 0.0;
    S[11] = amp_thr ? (double)amp_thr[b] : 0.0;
    S[12] = contact_thr ? (double)contact_thr[b] : 0.0;
    S[13] = bg_med ? (double)bg_med[b] : 0.0;
    S[14] = bad_count ? (double)bad_count[b] : 0.0;
    S[15] = 0.0;
}
void launch_fill_scalars(double *scalars, int nscal, const int *rel_count, const int *flipped, const float *amp_thr,
                         const float *contact_thr, const float *bg_med, const int *bad_count, int B, hipStream_t st)
{
    hipLaunchKernelGGL(k_fill_scalars, dim3((B + 63) / 64), dim3(64), 0, st, scalars, nscal, rel_count, flipped, amp_thr, contact_thr, bg_med,
                       bad_count, B);
}

// frames whose reliable mask is empty produce the upstream "return None" status and an all-NaN map
__global__ void k_mark_empty(const int *__restrict__ rel_count, int32_t *__restrict__ status, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && rel_count[b] == 0 && status[b] == 0) status[b] = 1;
}
void launch_mark_empty(const int *rel_count, int32_t *status, int B, hipStream_t st)
{
    hipLaunchKernelGGL(k_mark_empty, dim3((B + 63) / 64), dim3(64), 0, st, rel_count, status, B);
}

__global__ void k_copy_out(const float *__restrict__ depth, const uint8_t *__restrict__ reliable, const int32_t *__restrict__ status,
                           float *__restrict__ out_h, uint8_t *__restrict__ out_r, int P)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    size_t b = blockIdx.y;
    if (p >= P) return;
    size_t i = b * (size_t)P + p;
    bool empty = status[b] == 1 || status[b] == 3;     // empty reliable mask (upstream returns None) / no carrier (pair mode)
    if (out_h) out_h[i] = empty ? nanf32() : depth[i];
    if (out_r) out_r[i] = empty ? 0 : reliable[i];
}
void launch_copy_out(const float *depth, const uint8_t *reliable, const int32_t *status, float *out_h, uint8_t *out_r, int B, int P,
                     hipStream_t st)
{
    hipLaunchKernelGGL(k_copy_out, dim3((P + 255) / 256, B), dim3(256), 0, st, depth, reliable, status, out_h, out_r, P);
}

}  // namespace vf
