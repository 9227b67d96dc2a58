/*
 * vistaf_ftp.h -- C ABI of the MI355X-native (gfx950) VISTAF image -> height-map -> force path.
 *
 * Drop-in boundary for the reference's per-frame Fourier-Transform-Profilometry path.  The
 * reference (rimelq/VISTAF-RoboSkin-Vision-Integrated-Multimodal-Sensor) has no FFI: the path is a
 * plain Python call.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference tree):
 *
 *   vistaf_ftp_create / _set_reference   Code/shape_ftp.py:1467-1528, :1632-1639  (calibration load, ROI
 *                                        circle, apodisation, reference-frame demodulation) -- done once
 *                                        per session instead of once per frame (Code/height_to_force.py:384)
 *   vistaf_ftp_predict_batch             Code/shape_ftp.py:1428 `main(...)` steps :1641-2037 for B already
 *                                        aligned deformed crops, followed by the force tail
 *                                        Code/force_sensor.py:93-187 as called by
 *                                        Code/multimodal_sensor.py:388-419
 *   vistaf_depth_map_to_volume           Code/force_sensor.py:93-123 `depth_map_to_volume_cm3`
 *   vistaf_predict_force_from_volume     Code/force_sensor.py:149-167 `predict_force_from_volume`
 *
 * All image pointers are DEVICE pointers (HIP), planar, row-major, batch-major.  `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 on success or a negative VISTAF_E_* code;
 * no exception crosses the ABI.  vistaf_ftp_last_error() returns a thread-local message.
 */
#ifndef VISTAF_FTP_H
#define VISTAF_FTP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VISTAF_FTP_ABI_VERSION 2   /* 2: hole-stage constants in vistaf_ftp_config, vistaf_ftp_predict_pairs */

/* error codes */
#define VISTAF_OK 0
#define VISTAF_E_INVALID (-1)     /* bad argument (ValueError / KeyError upstream: shape_ftp.py:678,:700) */
#define VISTAF_E_HIP (-2)         /* HIP runtime failure */
#define VISTAF_E_STATE (-3)       /* predict before set_reference, batch > max_batch ... */
#define VISTAF_E_NOCARRIER (-4)   /* no carrier peak found in the reference frame spectrum */

/* per-frame status written by predict_batch (d_status[b]) */
#define VISTAF_FRAME_OK 0
#define VISTAF_FRAME_EMPTY_RELIABLE 1   /* upstream: main() logs and returns None (shape_ftp.py:1677-1679) */
#define VISTAF_FRAME_QUEUE_OVERFLOW 2   /* internal work queue exhausted (never for valid sizes) */
#define VISTAF_FRAME_NO_CARRIER 3       /* pair mode: no usable carrier peak in this sample's reference frame.  DELIBERATE RESTRICTION: a
                                         * carrier whose (2*patch_half_width_bins+1)^2 patch is clipped by the spectrum border (carrier
                                         * within 10 bins of Nyquist, i.e. a fringe period of ~2 px) is refused in pair mode, whereas
                                         * upstream clips the patch and goes on (shape_ftp.py:930-948); session mode (set_reference)
                                         * accepts clipped patches as upstream does.  Parity for that edge is unpinned (no stored
                                         * output of the reference covers it). */

/* input frame formats */
#define VISTAF_FMT_GRAY_U8 0    /* [B,h,w] uint8 (cv2.cvtColor(...,BGR2GRAY) already applied) */
#define VISTAF_FMT_BGR_U8 1     /* [B,h,w,3] uint8 interleaved, OpenCV channel order           */
#define VISTAF_FMT_GRAY_F16 2   /* [B,h,w] IEEE half holding integral values 0..255            */
#define VISTAF_FMT_BGR_F16 3    /* [B,h,w,3] IEEE half, interleaved                            */

/* calibration curve types (shape_ftp.py:682-700, force_sensor.py:129-167) */
#define VISTAF_CURVE_LINEAR0 0           /* a*v                       */
#define VISTAF_CURVE_LINEAR 1            /* a*v + b                   */
#define VISTAF_CURVE_POLY2 2             /* a*v*v + b*v + c  (c2,c1,c0) */
#define VISTAF_CURVE_SAT_EXP 3           /* a*(1-exp(-b*max(v,0)))    */
#define VISTAF_CURVE_GROWTH 4            /* a*(exp(b*max(v,0))-1)     */
#define VISTAF_CURVE_HINGE_SATURATING 5  /* a*((1-exp(-b*max(v-c,0)))-(1-exp(-b*max(-c,0)))) */

typedef struct vistaf_curve {
    int32_t type;
    int32_t reserved;
    double a, b, c;
} vistaf_curve;

/* Constants of Code/shape_ftp.py:23-218 and Code/force_sensor.py:33-34 (defaults = as shipped). */
typedef struct vistaf_ftp_config {
    int32_t patch_half_width_bins;      /* :27  10   */
    int32_t dc_exclusion;               /* :32  10   */
    int32_t fft_pad_px;                 /* :35  96   */
    int32_t roi_erode_px;               /* :86  0    */
    int32_t apod_taper_px;              /* :88  120  */
    int32_t reliable_edge_margin_px;    /* :93  6    */
    int32_t poly_order;                 /* :95  2    */
    int32_t frontier_zero_band_px;      /* :103 200  */
    int32_t valid_close_kernel;         /* :114 7    */
    int32_t valid_close_iters;          /* :115 1    */
    int32_t bad_pixel_enable;           /* :118 1    */
    int32_t bad_dilate_ksize;           /* :121 5    */
    int32_t bad_dilate_iters;           /* :122 1    */
    int32_t bad_inpaint_radius;         /* :123 3    */
    int32_t dilate_kernel_size;         /* :129 15   */
    int32_t dilate_iters;               /* :130 2    */
    int32_t n_fft_peaks;                /* :168 12   */
    int32_t plane_order_for_removal;    /* :212 1    */
    int32_t irls_iters;                 /* :1100 6   */
    int32_t hole_neighborhood_px;       /* :140 11   (only read when reliable_smooth_sigma_px == 0, see :1770-1801) */
    int32_t hole_min_dist_px;           /* :142 4    */
    int32_t inpaint_radius;             /* :144 5    */
    double pre_blur_sigma_px;           /* :38  1.5  */
    double amp_valid_percentile;        /* :90  25   */
    double quality_smooth_sigma_px;     /* :91  6    */
    double reliable_smooth_sigma_px;    /* :96  2.5  */
    double illum_sigma_px;              /* :110 45   */
    double bad_intensity_percentile;    /* :119 99.9 */
    double bad_gradient_percentile;     /* :120 99.7 */
    double contact_core_percentile;     /* :127 8    */
    double contact_percentile;          /* :128 92   */
    double min_contact_frac;            /* :131 0.002*/
    double max_contact_frac;            /* :132 0.40 */
    double unreliable_smooth_sigma_px;  /* :148 9    */
    double contact_blob_min_peak_mm;    /* :62  0.1  */
    double contact_blob_min_peak_rel_frac; /* :63 1/3 */
    double peak_max_dy_from_center;     /* :203 0.12 */
    double irls_c;                      /* :1100 4.685 */
    double grating_pitch_mm;            /* force_sensor.py:33  2.0  */
    double depth_eps_mm;                /* force_sensor.py:34  0.01 */
    double hole_known_fraction;         /* :141 0.70 */
} vistaf_ftp_config;

/* per-frame scalar record written by predict_batch: d_scalars[b*VISTAF_NSCALARS + i] (double) */
#define VISTAF_NSCALARS 16
#define VISTAF_S_VOLUME_CM3 0        /* force_sensor.py:118-122 */
#define VISTAF_S_CONTACT_AREA_MM2 1  /* :119 */
#define VISTAF_S_MAX_DEPTH_MM 2      /* :120 */
#define VISTAF_S_FORCE_N 3           /* :149-167 */
#define VISTAF_S_ARGMAX_DEPTH_INDEX 4/* shape_ftp.py:1945-1959: row-major index of max depth (mm), -1 if none */
#define VISTAF_S_PERIOD_PX 5         /* shape_ftp.py:2015-2027 */
#define VISTAF_S_MM_PER_PX 6         /* force_sensor.py:173-187 */
#define VISTAF_S_MIN_UNITLESS 7      /* phase_to_height.py:1009-1016 value  */
#define VISTAF_S_ARGMIN_UNITLESS_INDEX 8 /* phase_to_height.py:1009-1016 row-major index */
#define VISTAF_S_RELIABLE_COUNT 9
#define VISTAF_S_SIGN_FLIPPED 10     /* shape_ftp.py:1759-1768 */
#define VISTAF_S_AMP_THRESHOLD 11    /* shape_ftp.py:749 */
#define VISTAF_S_CONTACT_THRESHOLD 12/* shape_ftp.py:1720-1732 (the threshold finally used) */
#define VISTAF_S_BG_MEDIAN 13        /* shape_ftp.py:1746 */
#define VISTAF_S_BAD_PIXELS 14       /* shape_ftp.py:826 count */
#define VISTAF_S_RESERVED 15

/* reference-frame info returned by vistaf_ftp_get_reference_info: out[0..7] =
 * peak_x_refined, peak_y_refined, kx, ky, fft_h, fft_w, estimated_period_px, mm_per_px */
#define VISTAF_NREFINFO 8

typedef struct vistaf_ftp_handle vistaf_ftp_handle;

int vistaf_ftp_abi_version(void);
const char *vistaf_ftp_last_error(void);

/* Fill *cfg with the constants as shipped in Code/shape_ftp.py:23-218. */
int vistaf_ftp_default_config(vistaf_ftp_config *cfg);

/* Create a session for h x w crops with ROI circle (cx, cy, r) in crop coordinates
 * (shape_ftp.py:1514-1528), at most max_batch frames per predict call.  height_curve maps unitless
 * height -> mm (shape_ftp.py:672-705; use_negated_height = JSON "use_negated_height_for_fit"),
 * force_curve maps volume cm^3 -> N (force_sensor.py:149-167). */
int vistaf_ftp_create(const vistaf_ftp_config *cfg, int h, int w, int cx, int cy, int r, int max_batch,
                      const vistaf_curve *height_curve, int use_negated_height,
                      const vistaf_curve *force_curve, vistaf_ftp_handle **out);

/* Demodulate the reference frame on the GPU (carrier searched, shape_ftp.py:1632-1639) and cache its
 * state.  d_ref: one frame in `format`.  Synchronises `stream` (one-time setup). */
int vistaf_ftp_set_reference(vistaf_ftp_handle *hd, const void *d_ref, int format, void *stream);

int vistaf_ftp_get_reference_info(const vistaf_ftp_handle *hd, double *out8);

/* Process `batch` deformed frames (already aligned to the reference crop).  Outputs (device):
 *   d_height_mm [B,h,w] float32, NaN outside ROI      (main()'s "height_map_mm_crop",  :2031)
 *   d_reliable  [B,h,w] uint8 0/1                     ("output_reliable_crop",          :2033)
 *   d_scalars   [B,VISTAF_NSCALARS] double
 *   d_status    [B] int32 (VISTAF_FRAME_*)
 * Any output pointer may be NULL.  Asynchronous on `stream`. */
int vistaf_ftp_predict_batch(vistaf_ftp_handle *hd, const void *d_frames, int format, int batch,
                             float *d_height_mm, uint8_t *d_reliable, double *d_scalars,
                             int32_t *d_status, void *stream);

/* Uncached pairs: sample b = (d_refs[b], d_defs[b]), both in `format`; every sample's reference frame is demodulated with its own
 * carrier search and its deformed frame locked to that carrier -- what Code/height_to_force.py:384 does when it calls shape_ftp.main once
 * per image (shape_ftp.py:1632-1653).  Needs no vistaf_ftp_set_reference.  Outputs as vistaf_ftp_predict_batch; d_status[b] =
 * VISTAF_FRAME_NO_CARRIER when sample b's reference spectrum has no usable carrier peak.  Asynchronous on `stream`, except that the
 * first call and every call with a larger batch than any before allocate (and release) the carrier-search buffers, which synchronises. */
int vistaf_ftp_predict_pairs(vistaf_ftp_handle *hd, const void *d_refs, const void *d_defs, int format, int batch,
                             float *d_height_mm, uint8_t *d_reliable, double *d_scalars, int32_t *d_status, void *stream);

/* Reference-frame info of the samples of the last predict_pairs: out[b*VISTAF_NREFINFO + i], fields as
 * vistaf_ftp_get_reference_info.  Synchronises `stream`. */
int vistaf_ftp_get_pair_info(vistaf_ftp_handle *hd, int batch, double *out, void *stream);

/* Copy a named intermediate plane of the last predict_batch (parity tests / debugging) into d_dst.
 * Returns the number of bytes per frame through *bytes_per_frame; d_dst may be NULL to query. */
int vistaf_ftp_get_intermediate(vistaf_ftp_handle *hd, const char *name, void *d_dst, int batch,
                                size_t *bytes_per_frame, void *stream);

/* Per-stage device time of the last predict_batch in milliseconds (names via _stage_name). */
int vistaf_ftp_stage_count(void);
const char *vistaf_ftp_stage_name(int i);
int vistaf_ftp_enable_stage_timing(vistaf_ftp_handle *hd, int enable);
int vistaf_ftp_get_stage_times(vistaf_ftp_handle *hd, float *ms_out, int n);

void vistaf_ftp_destroy(vistaf_ftp_handle *hd);

/* force_sensor.py:93-123 on device maps: d_height [B,h,w] float32; d_roi [B,h,w] uint8 or NULL for
 * roi = isfinite(height) (multimodal_sensor.py:388).  d_out [B,3] double = volume_cm3, area_mm2,
 * max_depth_mm. */
int vistaf_depth_map_to_volume(const float *d_height, const uint8_t *d_roi, int batch, int h, int w,
                               double mm_per_px, double depth_eps_mm, double *d_out, void *stream);

/* force_sensor.py:149-167 (host scalar). Returns VISTAF_E_INVALID for an unknown curve type. */
int vistaf_predict_force_from_volume(const vistaf_curve *curve, double volume_cm3, double *force_out);

#ifdef __cplusplus
}
#endif
#endif /* VISTAF_FTP_H */
