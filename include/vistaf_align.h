/* vistaf_align.h -- C ABI of the pre-path alignment, part of libvistaf_ftp.so.
 *
 * Replaces, on the MI355X, the lines of the reference that turn two decoded photographs into the aligned ROI crops
 * the FTP path (vistaf_ftp.h) consumes -- Code/shape_ftp.py:1471-1537:
 *
 *   cv2.cvtColor(BGR2GRAY)                                             :1485-1486, :1514-1515
 *   estimate_global_shift (GaussianBlur 7, Hanning window, cv2.phaseCorrelate)   :529-535, :1488
 *   cv2.warpAffine(def_bgr, [[1,0,dx],[0,1,dy]], INTER_LINEAR, BORDER_REFLECT)   :1491-1494
 *   ROI bounding-box crop from the fixed circle                                  :1500-1512
 *   align_crop_ecc (GaussianBlur 5 on /255 floats, cv2.findTransformECC MOTION_EUCLIDEAN with the circular mask,
 *                   cv2.warpAffine(INTER_LINEAR | WARP_INVERSE_MAP, BORDER_REFLECT))  :549-578, :1528-1535
 *
 * Image decoding (cv2.imread) stays on the host.  All image pointers are HIP device pointers; `stream` is a hipStream_t
 * passed as void*.  Every function returns 0 or a negative VISTAF_E_* code (vistaf_ftp.h); vistaf_ftp_last_error() holds
 * the message.  The 2-D FFTs of the phase correlation are plain library transforms (hipFFT); everything else is
 * hand-written HIP.
 */
#ifndef VISTAF_ALIGN_H
#define VISTAF_ALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vistaf_align_handle vistaf_align_handle;

typedef struct vistaf_align_config {
    int32_t apply_global_shift;   /* APPLY_GLOBAL_SHIFT (shape_ftp.py:174), default 1 */
    int32_t use_ecc;              /* USE_ECC_CROP_ALIGNMENT (:176), default 1 */
    int32_t ecc_iters;            /* ECC_ITERS (:178), default 300 */
    int32_t gray_coeffs;          /* cv2.cvtColor(BGR2GRAY) fixed point: 0 = OpenCV 4.x, (B*3735 + G*19235 + R*9798 + 2^14) >> 15 (default);
                                   * 1 = OpenCV 3.x, (R*4899 + G*9617 + B*1868 + 2^13) >> 14.  The reference pins no version */
    double ecc_eps;               /* ECC_EPS (:179), default 1e-7 */
    double ecc_gauss_sigma;       /* ECC_GAUSS_FILT (:180), default 5 */
    double shift_blur_sigma;      /* the sigma of estimate_global_shift (:530-531), 7 */
} vistaf_align_config;

/* per-frame record written by vistaf_align_batch: doubles */
#define VISTAF_ALIGN_NINFO 12
#define VISTAF_AI_SHIFT_X 0       /* phaseCorrelate shift (dx, dy) and response (:534) */
#define VISTAF_AI_SHIFT_Y 1
#define VISTAF_AI_RESPONSE 2
#define VISTAF_AI_WARP 3          /* 6 entries: the 2x3 ECC warp, row-major (identity when ECC is off or failed) */
#define VISTAF_AI_RHO 9           /* ECC correlation coefficient (NaN when ECC failed: cv2.error upstream, :576-578) */
#define VISTAF_AI_ECC_ITERS 10    /* iterations executed */
#define VISTAF_AI_ECC_FAILED 11   /* 1: lambda_d <= 0 or NaN rho -> the unaligned crop is returned, as upstream does */

void vistaf_align_default_config(vistaf_align_config *cfg);

/* full-frame size H x W, fixed ROI circle (centre cx, cy and radius r in full-frame pixels: circle_from_3_points, :1499) */
int vistaf_align_create(const vistaf_align_config *cfg, int H, int W, int cx, int cy, int r, int max_batch, vistaf_align_handle **out);
void vistaf_align_destroy(vistaf_align_handle *h);

/* crop geometry (:1502-1519): crop box [x1,x2) x [y1,y2) in the full frame, crop size, circle in crop coordinates */
int vistaf_align_geometry(const vistaf_align_handle *h, int32_t *x1, int32_t *y1, int32_t *x2, int32_t *y2, int32_t *crop_h, int32_t *crop_w,
                          int32_t *cx_local, int32_t *cy_local, int32_t *r_local);

/* reference photograph [H,W,3] uint8 BGR: windowed spectrum for the phase correlation, grey ROI crop, ECC template.
 * d_ref_gray_crop (optional, [crop_h, crop_w] uint8) receives the reference crop for vistaf_ftp_set_reference. */
int vistaf_align_set_reference(vistaf_align_handle *h, const uint8_t *d_ref_bgr, uint8_t *d_ref_gray_crop, void *stream);

/* B deformed photographs [B,H,W,3] uint8 BGR -> aligned grey ROI crops [B,crop_h,crop_w] uint8 (input of
 * vistaf_ftp_predict_batch with VISTAF_FMT_GRAY_U8) and the [B, VISTAF_ALIGN_NINFO] records.  Synchronises the stream
 * (the ECC iteration count is data dependent). */
int vistaf_align_batch(vistaf_align_handle *h, const uint8_t *d_def_bgr, int B, uint8_t *d_def_gray_aligned, double *d_info, void *stream);

#ifdef __cplusplus
}
#endif
#endif
