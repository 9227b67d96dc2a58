// Launcher declarations for the gfx950 FTP kernels.  All pointers are device pointers; planes are
// [B, P] (P = h*w) unless noted; `st` is the HIP stream every launch goes to.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hpp"

namespace vf {

// Which of the parity-equivalent kernels a stage runs.  Defaults are the production kernels; the alternatives are fallbacks for
// sizes the defaults do not cover and stay selectable so the parity tests can pin them against the defaults (csrc/test_hooks.h).
struct Tiers {
    int inpaint = 2;            // 2: frame-window march (k_telea_window) + whole-frame fallback; 1: whole-frame kernel only; 0: cluster front end first
    int flood = 2;              // 2: batched pops (k_unwrap_flood_batch / k_unwrap_flood_big); 1: one pop per step (k_unwrap_flood_hot); 0: frontier scan; 3: test only (test_hooks.h)
    int chamfer_twopass = 0;    // 1: force the one-wave two-pass chamfer even where the LDS closed form applies
    int fit_capped = 0;        // 0: 128-VGPR column polyfit (default since the march and the flood stopped pinning CUs for milliseconds: 1.49 against 1.79 ms per step); 1: register-capped variant (96 VGPRs, shares a CU with LDS-heavy one-wave kernels)
    int telea_two_tier = 1;     // 1: 111 KB first tier of the window march + full-size retry of the frames it hands back; 0: full size only
    int unwrap_fast = 1;        // 1: frames whose wrapped field is verified path-independent skip the priority flood (k_unwrap_fast.hip); 0: always flood
    int big_chain = 1;          // 1: frames of 512 x 512 and more take k_big.hip's chains of streaming kernels for the exact selections and the IRLS fits; 0: one workgroup per frame
    int big_queue_lds = 1;      // 1: a big cluster's march keeps its queue in LDS whenever its cell counts bound the queue (k_inpaint_big.hip); 0: always this wave's slice of global memory
    int telea_mw = 1;           // 1: the 16-wave window kernel (ordering pass + dataflow fills, k_inpaint_mw.hip) as first tier, single-wave tiers behind it; 0: single-wave tiers only
};

struct RowSpanSE {      // structuring element as per-row x spans (cv::getStructuringElement ELLIPSE)
    int k;              // k x k, anchor at centre (k <= 33)
    int8_t lo[33];      // relative x offset of first set element in row i (lo > hi: empty row)
    int8_t hi[33];
};

// ---- k_basic.hip ------------------------------------------------------------------------------
void launch_to_gray(const void *frames, int format, float *gray, int B, int P, hipStream_t st);
void launch_sobel_mag(const float *img, float *grad, int B, int h, int w, hipStream_t st);
void launch_bad_flags(const float *img, const float *grad, const uint8_t *valid, const float *thr_hi, const float *thr_g,
                      uint8_t *bad, int B, int P, hipStream_t st);
void launch_morph(const uint8_t *src, uint8_t *dst, int B, int h, int w, const RowSpanSE &se, bool dilate,
                  const uint8_t *and_static, const uint8_t *and_frame, hipStream_t st, uint16_t *prefix_scratch = nullptr);
void launch_morph_seq(const uint8_t *src, uint8_t *dst, uint8_t *tmp, int B, int h, int w, const RowSpanSE &se, const int *dilates, int n,
                      const uint8_t *and_static, const uint8_t *and_frame, hipStream_t st, uint16_t *prefix_scratch = nullptr);
void launch_gauss_rows(const float *src, float *dst, const float *kern, int ksize, int B, int h, int w, hipStream_t st);
void launch_gauss_blur(const float *src, float *tmp, float *dst, const float *kern, int ksize, int B, int h, int w, hipStream_t st);
void launch_gauss_cols(const float *src, float *dst, const float *kern, int ksize, int B, int h, int w, hipStream_t st);
void launch_illum_norm(const float *img, const float *blur, float *out, int B, int P, hipStream_t st);
void launch_mul_static(const float *a, const float *stat, float *out, int B, int P, hipStream_t st);
void launch_count_u8(const uint8_t *m, int *counts, int B, int P, hipStream_t st);

// ---- k_select.hip -----------------------------------------------------------------------------
// Per frame: values vals[b*P+i] (|.| if use_abs) over pixels with mask != 0 (mask_stride 0: one static
// mask for all frames), finite, and (le_thr ? value <= le_thr[b] : true).
// reqs[j] >= 0: percentile with q32 = reqs[j];  reqs[j] < 0: median.   out[b*nreq+j], counts[b].
void launch_select(const float *vals, const uint8_t *mask, size_t mask_stride, const float *le_thr, bool use_abs,
                   const float *reqs_dev, int nreq, float *out, int *counts, int B, int P, hipStream_t st, void *big_scratch = nullptr);      // big_scratch: k_big.hip's chain for large frames

// ---- k_dft.hip / k_dft_tables.hip ----------------------------------------------------------------
// carrier of one reference frame (shape_ftp.py:878-913, :930-961)
struct CarrierGeom {
    double peak_x, peak_y;      // refined peak (fftshift layout)
    double kx, ky;              // carrier in bins
    double dpx, dpy;            // sub-bin remainder handled by the ramp (0 when both <= 1e-6)
    double period;              // Wf / |kx| (0: invalid)
    int x0, y0, ph, pw;         // patch origin and size in the shifted spectrum
    int px_i, py_i;
    int px_raw, py_raw;         // the chosen integer peak itself (before the sub-bin refinement)
    int ok;
    int keep_carrier;           // 0: the patch is re-centred at DC before the inverse transform (FTP, :945-948); 1: band-pass in place
};
// table strides are in elements per frame (0: one table for the whole batch)
void launch_dft_forward(const float *iw, const float *mu, const double2 *Ex, const double2 *Ey, size_t tab_stride_x, size_t tab_stride_y,
                        const float *win, double2 *tmpT, double2 *patch, int patch_stride, int B, int h, int w, int ph, int pw, hipStream_t st);
// field (may be null): float64 field out; cref/amp_ref (may be null): reference field -> wrapped phase difference and amp product
void launch_dft_inverse(const double2 *patch, int patch_stride, const double2 *Gx, const double2 *Gy, size_t tab_stride_x, size_t tab_stride_y,
                        double2 *tmpQ, double2 *field, float *amp, const double2 *cref, const float *amp_ref, size_t ref_stride, float *prod,
                        float *wrapped, int B, int h, int w, int ph, int pw, hipStream_t st);
void launch_dft_full_mag(const float *iw, const float *mu, const double2 *Ex_full, const double2 *Ey_full, double2 *tmp,
                         double *mag, int B, int h, int w, int Hf, int Wf, hipStream_t st);
void launch_top_peaks(const double *mag, int B, int Hf, int Wf, int dc, int npeaks, double *out_xyv /* [B][192] */, hipStream_t st);
void launch_carrier_choose(const double *peaks, int npk, const double *mag, int Hf, int Wf, int bw, double max_dy_frac, CarrierGeom *geom, int B,
                           hipStream_t st);
void launch_build_tables(const CarrierGeom *geom, int geom_stride, double2 *Ex, double2 *Ey, double2 *Gx, double2 *Gy, size_t stride_x,
                         size_t stride_y, int B, int h, int w, int pad, int Hf, int Wf, int pmax, hipStream_t st);
// pair mode: frames without a usable carrier (or whose patch is clipped by the spectrum border) get status VISTAF_FRAME_NO_CARRIER
void launch_pair_status(const CarrierGeom *geom, int pmax, int32_t *status, const int32_t *status2, int B, hipStream_t st);
void launch_build_full_tables(double2 *Exf, double2 *Eyf, int h, int w, int pad, int Hf, int Wf, hipStream_t st);

// ---- k_cc_dist.hip ----------------------------------------------------------------------------
void launch_threshold_mask(const float *q, const uint8_t *roi, const float *thr, uint8_t *out, int B, int P, hipStream_t st);
void launch_cc_label(const uint8_t *mask, int32_t *labels, int B, int h, int w, hipStream_t st);
void launch_cc_largest(const int32_t *labels, int32_t *area_scratch, unsigned long long *best, const uint8_t *and_static,
                       uint8_t *out, int B, int P, hipStream_t st);
void launch_chamfer(const uint8_t *src, bool invert, int32_t *rowdist, float *dist, int B, int h, int w, int cap_px, hipStream_t st,
                    bool force_twopass = false);
void launch_chamfer_pair(const uint8_t *src, int32_t *tmp_a, float *dist_a, int32_t *tmp_b, float *dist_b, int B, int h, int w, int cap_px,
                         hipStream_t st, bool force_twopass = false);
void launch_erode_by_dist(const float *dist, const uint8_t *src, float margin, uint8_t *out, int B, int P, hipStream_t st);

// ---- k_inpaint.hip ----------------------------------------------------------------------------
size_t inpaint_scratch_bytes_per_frame(int h, int w);
// `only` (device, [B], may be null): process just the frames with only[b] != 0
void launch_inpaint_telea(float *img, const uint8_t *bad, int range, void *scratch, int32_t *status, const int32_t *only, int B, int h, int w,
                          hipStream_t st, bool round_u8 = false);      // round_u8: 8-bit image semantics (values 0..255 held as floats, OpenCV's rounding of every estimate)

// ---- k_inpaint_win.hip (LDS-resident window kernel; returns the per-frame fallback flags for launch_inpaint_telea)
size_t inpaint_win_scratch_bytes(int B);
// ---- k_inpaint_mw.hip (16 waves per frame: ordering pass, then the estimates as a dataflow; flags the frames it cannot take in fb[])
bool inpaint_window_mw_supported(int range);
void launch_telea_window_mw(float *img, const uint8_t *bad, const int32_t *box, int32_t *fb, int range, int B, int h, int w, hipStream_t st);
int32_t *launch_inpaint_window(float *img, const uint8_t *bad, int range, void *scratch, int B, int h, int w, hipStream_t st,
                               hipEvent_t ev_march = nullptr, bool two_tier = true, bool mw = true);

// ---- k_inpaint_cl.hip (cluster-parallel front end; leaves oversized clusters in *bad_big_out) ----------
size_t inpaint_cl_scratch_bytes_per_frame(int h, int w);
bool inpaint_clusters_supported(int range);
struct ClusterPlanes { const int32_t *labels, *list, *count, *xmin, *ymin, *xmax, *ymax; const uint8_t *dil; };     // [B, P] planes indexed by component root; dil = hole mask dilated by range + 1
// the clusters the LDS windows left over, each on its own wave over padded global planes with the queue in LDS (k_inpaint_big.hip)
size_t inpaint_big_scratch_bytes_per_frame(int h, int w);
bool inpaint_big_supported(int range);
void launch_inpaint_big_clusters(float *img, const uint8_t *bad_big, int range, void *scratch, int32_t *status, const ClusterPlanes &left, int B, int h,
                                 int w, hipStream_t st, bool lds_queue = true);
void launch_inpaint_clusters(float *img, const uint8_t *bad, int range, void *scratch, uint8_t **bad_big_out, ClusterPlanes *left, int B, int h, int w,
                             hipStream_t st);

// ---- k_unwrap.hip -----------------------------------------------------------------------------
size_t unwrap_scratch_bytes_per_frame(int h, int w);
void launch_unwrap(const float *wrapped, const float *quality, const uint8_t *mask, float *unwrapped, int32_t *parent,
                   void *scratch, int32_t *status, int B, int h, int w, hipStream_t st, hipEvent_t ev_mid,
                   hipEvent_t ev_flood = nullptr, int flood_tier = 2, int32_t *need_buf = nullptr);      // need_buf [B]: enables the consistency check (k_unwrap_fast.hip)

// ---- k_fit.hip --------------------------------------------------------------------------------
// min_count: fitted (mask & finite) pixels needed (:1103); min_mask_count: mask pixels needed, NaN included (debug_ramp's own gate, :1364)
void launch_robust_polyfit(const float *z, const uint8_t *mask, int order, int iters, float c, int min_count, int min_mask_count, float *coef_out,
                           float *resid_out, int B, int h, int w, hipStream_t st, int capped = 1, void *big_scratch = nullptr);

// ---- k_holes.hip (hole stage, shape_ftp.py:1153-1204, :1770-1801; live only when reliable_smooth_sigma_px == 0) ----------------------
void launch_zeroed_keep_nan(const float *detr, const float *bg_med, const uint8_t *reliable, float *hmap, int B, int P, hipStream_t st);
void launch_hole_candidates(const float *hmap, const uint8_t *reliable, const float *dist, int ksize, float frac_thr, float min_dist, uint8_t *cand,
                            int B, int h, int w, hipStream_t st);
void launch_hole_tmp(const float *hmap, const uint8_t *reliable, const uint8_t *cand, const float *med, float *tmp, int B, int P, hipStream_t st);
void launch_hole_zin(float *tmp_zin, const float *fill, int B, int P, hipStream_t st);
void launch_hole_merge(float *hmap, const uint8_t *reliable, const uint8_t *cand, const float *zin, uint8_t *out_rel, int B, int P, hipStream_t st);

// ---- k_lab.hip (temperature modality: feature planes and colour support, temperature_sensor.py:278-293, :790-799) --------------------
constexpr int LAB_SHIFT = 12, LAB_SHIFT2 = 15, LAB_GAMMA_SHIFT = 3, LAB_CBRT_N = 256 * 3 / 2 * (1 << LAB_GAMMA_SHIFT);
struct LabCoef { int c[9]; int lscale, lshift; };     // sRGB -> XYZ / white point in 2^12 fixed point, rows X, Y, Z, columns R, G, B
void launch_feature_planes(const uint8_t *bgr, const uint16_t *gamma_tab, const uint16_t *cbrt_tab, const LabCoef &cf, bool blur, float *L, float *a,
                           float *b, float *gray, int H, int W, hipStream_t st);
void launch_color_support(const float *a, const float *b, const uint8_t *light_d, const uint8_t *roi_eff, const uint8_t *sat, float chroma_min,
                          float *chroma, uint8_t *support, size_t n, hipStream_t st);

// ---- k_post.hip -------------------------------------------------------------------------------
struct PostParams {
    double mm_per_px, depth_eps_mm, period_px;
    Curve force_curve;
    const CarrierGeom *pair_geom = nullptr;     // pair mode: per-frame period (mm_per_px = grating_pitch_mm / period, force_sensor.py:173-187)
    double grating_pitch_mm = 0.0;
};
void launch_contact_mask(const float *res, const uint8_t *reliable, const float *thr3, const int *rel_count, int *contact_count,
                         float min_frac, float max_frac, uint8_t *contact, float *thr_used, int B, int P, hipStream_t st);
void launch_background(const uint8_t *reliable, const uint8_t *contact_d, const int *rel_count, int *bg_count, uint8_t *background,
                       int B, int P, hipStream_t st);
void launch_sub_scalar_mask(const float *src, const float *scalar, const uint8_t *mask, float *z0, float *m_out, int B, int P,
                            hipStream_t st);
void launch_div_planes(const float *num, const float *den, float *out, int B, int P, hipStream_t st);
void launch_core_flip(float *hmap, const float *core_med, int *flipped, int B, int P, hipStream_t st);
void launch_frontier_compose(const float *hmap, const uint8_t *reliable, const uint8_t *roi, const float *dist_in, float band,
                             float *hfinal_z0, int B, int P, hipStream_t st);
void launch_finalize_unitless(const float *hfinal_z0, const float *smooth_num, const float *roi_den, const uint8_t *reliable,
                              const uint8_t *roi, const float *dist_out, float band, int use_band, float *unitless, int B, int P,
                              hipStream_t st);
void launch_to_mm(const float *unitless, const uint8_t *roi, Curve curve, int use_neg, float *depth, uint8_t *cand,
                  unsigned int *gmax_bits, int B, int P, hipStream_t st);
void launch_blob_filter(float *depth, const uint8_t *cand, const int32_t *labels, unsigned int *peak_bits,
                        const unsigned int *gmax_bits, float min_peak_mm, double rel_frac, uint8_t *kept, int B, int P,
                        hipStream_t st);
void launch_tail(const float *height_mm, const uint8_t *roi_or_null, const float *unitless_or_null, const uint8_t *roi_static,
                 PostParams pp, double *scalars, int nscal, double *out3_or_null, int B, int P, hipStream_t st, void *scratch = nullptr,
                 size_t scratch_bytes = 0);
void launch_fill_scalars(double *scalars, int nscal, const int *rel_count, const int *flipped, const float *amp_thr,
                         const float *contact_thr, const float *bg_med, const int *bad_count, int B, hipStream_t st);
void launch_mark_empty(const int *rel_count, int32_t *status, int B, hipStream_t st);
void launch_copy_out(const float *depth, const uint8_t *reliable, const int32_t *status, float *out_h, uint8_t *out_r, int B, int P,
                     hipStream_t st);

// ---- k_big.hip (large frames: selection and IRLS fit as chains of streaming kernels over all pixels of the batch)
size_t big_scratch_bytes(int B, int h, int w);
bool big_frames(int B, int P);
void launch_select_big(const float *vals, const uint8_t *mask, size_t mask_stride, const float *le_thr, bool use_abs, const float *reqs_dev, int nreq,
                       float *out, int *counts, int B, int P, void *scratch, hipStream_t st);
void launch_robust_polyfit_big(const float *z, const uint8_t *mask, int order, int iters, float c, int min_count, int min_mask_count, float *coef_out,
                               float *resid_out, int B, int h, int w, void *scratch, hipStream_t st);

// ---- k_tempmap.hip (map-domain stages of the temperature modality; parity unpinned, see the file)
struct TmAff { double m[6]; };          // source = M * (x, y, 1): the inverse map cv::warpAffine iterates with
struct TmFuse { float color_lo, color_hi, low_th, high_th, final_lo, final_hi; };
void launch_tm_clamp(const float *m, const uint8_t *roi, float lo, float hi, float *out, size_t P, hipStream_t st);
void launch_tm_stats(const float *m, const uint8_t *roi, uint32_t *stats, size_t P, hipStream_t st);
void launch_tm_scale(const float *m, const uint8_t *roi, const uint32_t *stats, float *scaled, uint8_t *miss, size_t P, hipStream_t st);
void launch_tm_unscale(const float *m, const uint8_t *roi, const uint32_t *stats, const float *filled, float *out, size_t P, hipStream_t st);
void launch_tm_fuse(const uint8_t *roi, const float *wide, const float *color, const TmFuse &c, float *fin, uint8_t *source, unsigned long long *counts, size_t P,
                    hipStream_t st);
void launch_tm_zero_nonfinite(const float *m, float *out, size_t P, hipStream_t st);
void launch_tm_warp_linear(const float *src, float *dst, const TmAff &a, int h, int w, hipStream_t st);
void launch_tm_warp_nearest(const uint8_t *src, uint8_t *dst, const TmAff &a, int h, int w, hipStream_t st);
void launch_tm_mask_nan(const float *m, const uint8_t *keep, float *out, size_t P, hipStream_t st);

}  // namespace vf
