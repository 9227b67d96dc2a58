"""CPU oracle for the first slice of the temperature modality.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/Code/temperature_sensor.py's periodic-stripe segmentation -- the step that splits the thermochromic grating into its
dark and light stripes before any temperature regression runs:

  circle_from_three_points / roi_mask_from_circle  :156-183
  bbox_from_mask / crop2d                           :194-216
  _find_top_peaks / _choose_carrier_peak            :316-360
  _illum_normalize                                  :363-375
  _make_saturation_mask                             :378-387
  _postprocess_mask                                 :390-406
  segment_dark_light_gratings_periodic_fft          :437-540
  compute_feature_planes                            :278-293   (cv2.GaussianBlur on uint8 = OpenCV's fixed-point path; cv2.cvtColor BGR2LAB
                                                                on uint8 = OpenCV's integer table path RGB2Lab_b; BGR2GRAY)
  dilate_bool_mask                                  :583-590
  chroma and the COLOR support mask of main()       :790-797

NumPy calls are the reference's; OpenCV calls (cvtColor BGR2GRAY, GaussianBlur, getStructuringElement ELLIPSE / RECT, dilate, morphologyEx
CLOSE / OPEN) go through oracle/cvlite.c / oracle/align_oracle.py like everywhere else in this oracle.  FFT precision follows
ftp_oracle.FFT_COMPLEX128 (see there).

Pinned by the masks the reference itself stored for its five demo photographs
(Multimodal_Sensor/Demos_report/<name>/temperature_sensing/mask_{roi,roi_eff,sat,dark,light}.png, written by temperature_sensor.py:803-812):
tests/golden/make_temp_seg_report.py -> tests/golden/temp_seg_report.json, and the FINAL_E masks as a committed fixture.
The temperature REGRESSORS (Huber pipelines stored as .joblib pickles) are out of reach of this build: pickles are not loadable under its
rules and the equations_*.txt files list coefficients in standardised variables without the scaler.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Tuple

import math

import numpy as np

from . import align_oracle as A
from . import cvlite as cv
from . import ftp_oracle as O


@dataclasses.dataclass
class TempSegConfig:
    """temperature_sensor.py:36-82 (as shipped)"""
    outer_circle: Tuple[Tuple[int, int], Tuple[int, int], Tuple[int, int]] = ((1845, 1818), (1517, 623), (2687, 914))   # :37-39
    crop_pad_px: int = 10                 # :49
    seg_band_radius: int = 22             # :67
    seg_dc_exclusion: int = 28            # :68
    seg_peak_max_dy_from_center: float = 0.14   # :71
    seg_illum_sigma: int = 20             # :72
    sat_thresh_gray: int = 245            # :75
    sat_dilate_ksize: int = 13            # :76
    post_close_kx: int = 3                # :79
    post_close_ky: int = 31               # :80
    post_open_kx: int = 3                 # :81
    post_open_ky: int = 7                 # :82
    n_peaks: int = 16                     # :457
    blur_ksize: int = 5                   # :52  BLUR_KSIZE (feature smoothing)
    color_chroma_min: float = 10.0        # :86  COLOR_CHROMA_MIN
    color_support_dilate: int = 3         # :87  COLOR_SUPPORT_DILATE


def circle_from_three_points(p1, p2, p3, eps: float = 1e-12):
    """:156-177"""
    x1, y1 = map(float, p1)
    x2, y2 = map(float, p2)
    x3, y3 = map(float, p3)
    a, b, c, d = x1 - x2, y1 - y2, x1 - x3, y1 - y3
    e = (x1 ** 2 - x2 ** 2 + y1 ** 2 - y2 ** 2) / 2.0
    f = (x1 ** 2 - x3 ** 2 + y1 ** 2 - y3 ** 2) / 2.0
    det = a * d - b * c
    if abs(det) < eps:
        raise RuntimeError("Cannot define circle: points are collinear (or nearly collinear).")
    cx = (d * e - b * f) / det
    cy = (-c * e + a * f) / det
    return float(cx), float(cy), float(np.hypot(x1 - cx, y1 - cy))


def roi_mask_from_circle(h: int, w: int, p1, p2, p3) -> np.ndarray:
    """:180-184"""
    cx, cy, r = circle_from_three_points(p1, p2, p3)
    yy, xx = np.ogrid[:h, :w]
    return (xx - cx) ** 2 + (yy - cy) ** 2 <= r ** 2


def bbox_from_mask(mask: np.ndarray, pad: int = 0):
    """:194-208: (y0, y1, x0, x1), end exclusive"""
    h, w = mask.shape[:2]
    ys, xs = np.where(mask)
    if ys.size == 0:
        return 0, h, 0, w
    return (int(max(0, ys.min() - int(pad))), int(min(h, ys.max() + int(pad) + 1)), int(max(0, xs.min() - int(pad))),
            int(min(w, xs.max() + int(pad) + 1)))


def _ensure_odd(k: int) -> int:
    k = int(k)
    return k if k % 2 == 1 else k + 1


def _rect_se(kx: int, ky: int) -> np.ndarray:
    """cv2.getStructuringElement(MORPH_RECT, (kx, ky)) embedded in the square element cvlite's morphology takes (anchor = centre both ways)"""
    k = max(kx, ky)
    se = np.zeros((k, k), np.uint8)
    se[(k - ky) // 2:(k - ky) // 2 + ky, (k - kx) // 2:(k - kx) // 2 + kx] = 1
    return se


def make_saturation_mask(gray_u8: np.ndarray, roi: np.ndarray, cfg: TempSegConfig) -> np.ndarray:
    """:378-387"""
    sat = (gray_u8 >= int(cfg.sat_thresh_gray)) & roi
    k = _ensure_odd(cfg.sat_dilate_ksize)
    if k > 1 and np.any(sat):
        sat = (cv.dilate(sat.astype(np.uint8) * 255, cv.ellipse_se(k), 1) > 127) & roi
    return sat


def illum_normalize(gray_f: np.ndarray, roi: np.ndarray, sigma: int) -> np.ndarray:
    """:363-375"""
    g = gray_f.astype(np.float32)
    if sigma is None or int(sigma) <= 0:
        mu = float(np.mean(g[roi])) if np.any(roi) else float(np.mean(g))
        mu = mu if abs(mu) > 1e-9 else 1.0
        return (g / mu).astype(np.float32)
    blur = cv.gaussian_blur(g, float(sigma))
    blur[blur < 1e-6] = 1.0
    norm = g / blur
    mu = float(np.mean(norm[roi])) if np.any(roi) else float(np.mean(norm))
    mu = mu if abs(mu) > 1e-9 else 1.0
    return (norm / mu).astype(np.float32)


def postprocess_mask(m: np.ndarray, roi: np.ndarray, cfg: TempSegConfig) -> np.ndarray:
    """:390-406: close (kx x ky rectangle) then open, inside roi"""
    if not np.any(m):
        return m
    k_close = _rect_se(_ensure_odd(max(1, cfg.post_close_kx)), _ensure_odd(max(1, cfg.post_close_ky)))
    k_open = _rect_se(_ensure_odd(max(1, cfg.post_open_kx)), _ensure_odd(max(1, cfg.post_open_ky)))
    mu8 = m.astype(np.uint8) * 255
    mu8 = cv.erode(cv.dilate(mu8, k_close, 1), k_close, 1)        # MORPH_CLOSE
    mu8 = cv.dilate(cv.erode(mu8, k_open, 1), k_open, 1)          # MORPH_OPEN
    return (mu8 > 127) & roi


def choose_carrier_peak(peaks, h: int, w: int, max_dy_frac: float):
    """:339-360 (SEG_FORCE_RIGHT_HALF_PLANE and SEG_PREFER_PEAK_NEAR_CENTER_ROW both True)"""
    return O.choose_carrier_peak(peaks, h, w, max_dy_frac)


def segment_dark_light_gratings_periodic_fft(image_bgr: np.ndarray, roi_full: np.ndarray, cfg: TempSegConfig = TempSegConfig()):
    """:437-540 -> (dark_final, light_final, pack)"""
    h, w = image_bgr.shape[:2]
    gray_u8 = A.bgr2gray_u8(image_bgr)
    gray = gray_u8.astype(np.float32)
    sat = make_saturation_mask(gray_u8, roi_full, cfg)
    roi_eff = roi_full & (~sat)
    if not np.any(roi_eff):
        raise RuntimeError("ROI became empty after saturation exclusion. Lower SAT_THRESH_GRAY / dilation.")
    g = gray.copy()
    med = float(np.median(g[roi_eff]))
    g[~roi_full] = med
    i_norm = illum_normalize(g, roi_eff, cfg.seg_illum_sigma)
    F = np.fft.fft2(i_norm.astype(np.float64) if O.FFT_COMPLEX128 else i_norm)
    F_shift = np.fft.fftshift(F)
    mag = np.abs(F_shift)
    peaks = O.find_top_peaks(mag, dc_exclusion=int(cfg.seg_dc_exclusion), n_peaks=cfg.n_peaks)
    if not peaks:
        raise RuntimeError("Could not find FFT peaks for stripe carrier.")
    peak_x, peak_y = choose_carrier_peak(peaks, h, w, cfg.seg_peak_max_dy_from_center)
    yy, xx = np.ogrid[:h, :w]
    bp = (xx - peak_x) ** 2 + (yy - peak_y) ** 2 <= float(cfg.seg_band_radius) ** 2
    z = np.fft.ifft2(np.fft.ifftshift(F_shift * bp))
    m = (i_norm - 1.0).astype(np.float32)
    c = np.sum(z[roi_eff] * m[roi_eff])
    phi0 = float(np.angle(c)) if np.isfinite(c) else 0.0
    z_rot = z * np.exp(-1j * phi0)
    s = np.real(z_rot).astype(np.float32)
    mask_a = (s >= 0) & roi_eff
    mask_b = (s < 0) & roi_eff
    mean_a = float(np.mean(gray[mask_a])) if np.any(mask_a) else 1e9
    mean_b = float(np.mean(gray[mask_b])) if np.any(mask_b) else 1e9
    if mean_a <= mean_b:
        dark, light, chosen = mask_a, mask_b, "A_is_dark"
    else:
        dark, light, chosen = mask_b, mask_a, "B_is_dark"
    raw_dark = dark
    dark = postprocess_mask(dark, roi_eff, cfg)
    dark_final = dark & roi_eff
    light_final = roi_eff & (~dark_final)
    cy, cx = h // 2, w // 2
    dx, dy = float(peak_x - cx), float(peak_y - cy)
    fmag = float(np.hypot(dx / float(w), dy / float(h)))
    dbg: Dict[str, Any] = {
        "peak_x": int(peak_x), "peak_y": int(peak_y), "phi0_rad": float(phi0), "mean_gray_A": mean_a, "mean_gray_B": mean_b, "chosen": chosen,
        "roi_pixels": int(np.count_nonzero(roi_full)), "roi_eff_pixels": int(np.count_nonzero(roi_eff)), "sat_pixels": int(np.count_nonzero(sat)),
        "dark_pixels": int(np.count_nonzero(dark_final)), "light_pixels": int(np.count_nonzero(light_final)),
        "carrier_angle_rad": float(np.arctan2(dy, dx)), "carrier_period_px": (1.0 / fmag) if fmag > 1e-9 else float("nan"),
    }
    pack = {"dbg": dbg, "fft_mag": mag, "signal": s, "roi_eff": roi_eff, "sat": sat, "peak": (peak_x, peak_y), "angle_rad": dbg["carrier_angle_rad"],
            "period_px": dbg["carrier_period_px"], "raw_dark": raw_dark, "i_norm": i_norm, "z": z}
    return dark_final, light_final, pack


# ---------------------------------------------------------------------------------------------
# feature planes and colour support (second slice): temperature_sensor.py:278-293, :583-590, :790-797
# ---------------------------------------------------------------------------------------------
def gaussian_blur_u8_ksize5(img_u8: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(uint8 image, (5, 5), 0).  For 8-bit images OpenCV runs its fixed-point filter: the kernel of ksize 5 / sigma 0 is the
    tabulated [1, 4, 6, 4, 1] / 16 (cv::getGaussianKernel's small_gaussian_tab), both passes are exact in 8.8 / 16.16 fixed point and the
    ONE rounding at the end is (sum + 128) >> 8; BORDER_REFLECT_101."""
    wts = np.array([1, 4, 6, 4, 1], np.int64)
    img = img_u8 if img_u8.ndim == 3 else img_u8[..., None]
    h, w = img.shape[:2]
    p = np.pad(img.astype(np.int64), ((2, 2), (2, 2), (0, 0)), mode="reflect")
    acc = np.zeros(img.shape, np.int64)
    for i in range(5):
        for j in range(5):
            acc += wts[i] * wts[j] * p[i:i + h, j:j + w]
    out = ((acc + 128) >> 8).astype(np.uint8)
    return out if img_u8.ndim == 3 else out[..., 0]


_LAB_SHIFT, _LAB_SHIFT2, _GAMMA_SHIFT = 12, 15, 3


def lab_tables_u8():
    """The integer tables of cv::RGB2Lab_b (color_lab.cpp): sRGB gamma (256 entries, scaled by 255 * 8), cube root (3072 entries, scaled by
    2^15) and the sRGB -> XYZ (D65) matrix divided by the white point, scaled by 2^12."""
    x = (np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float64)
    g = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    gtab = np.rint(255.0 * (1 << _GAMMA_SHIFT) * g).astype(np.int64)
    n = 256 * 3 // 2 * (1 << _GAMMA_SHIFT)
    xx = np.arange(n, dtype=np.float64) / (255.0 * (1 << _GAMMA_SHIFT))
    f = np.where(xx < 0.008856, xx * 7.787 + 0.13793103448275862, np.cbrt(xx))
    ctab = np.rint((1 << _LAB_SHIFT2) * f).astype(np.int64)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    wp = np.array([0.950456, 1.0, 1.088754])
    coef = np.rint((1 << _LAB_SHIFT) * m / wp[:, None]).astype(np.int64)
    return gtab, ctab, coef


def bgr2lab_u8(img_bgr: np.ndarray):
    """cv2.cvtColor(uint8 BGR, COLOR_BGR2LAB) -> (L, a, b) uint8 planes (8-bit Lab: L * 255 / 100, a + 128, b + 128)"""
    gtab, ctab, c = lab_tables_u8()
    b_, g_, r_ = gtab[img_bgr[..., 0]], gtab[img_bgr[..., 1]], gtab[img_bgr[..., 2]]

    def desc(v, s):
        return (v + (1 << (s - 1))) >> s
    fx = ctab[desc(r_ * c[0, 0] + g_ * c[0, 1] + b_ * c[0, 2], _LAB_SHIFT)]
    fy = ctab[desc(r_ * c[1, 0] + g_ * c[1, 1] + b_ * c[1, 2], _LAB_SHIFT)]
    fz = ctab[desc(r_ * c[2, 0] + g_ * c[2, 1] + b_ * c[2, 2], _LAB_SHIFT)]
    lscale = (116 * 255 + 50) // 100
    lshift = -((16 * 255 * (1 << _LAB_SHIFT2) + 50) // 100)
    L = desc(lscale * fy + lshift, _LAB_SHIFT2)
    a = desc(500 * (fx - fy) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    b = desc(200 * (fy - fz) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    return tuple(np.clip(v, 0, 255).astype(np.uint8) for v in (L, a, b))


def compute_feature_planes(image_bgr: np.ndarray, blur_ksize: int = 5) -> Dict[str, np.ndarray]:
    """:278-293 (blur_ksize 5, the shipped BLUR_KSIZE, or <= 1 for none)"""
    k = _ensure_odd(blur_ksize) if blur_ksize > 1 else 1
    if k > 1:
        if k != 5:
            raise NotImplementedError("only the shipped BLUR_KSIZE = 5 is restated")
        image_bgr = gaussian_blur_u8_ksize5(image_bgr)
    L, a, b = bgr2lab_u8(image_bgr)
    gray = A.bgr2gray_u8(image_bgr)
    return {"L": L.astype(np.float32), "a": a.astype(np.float32), "b": b.astype(np.float32), "gray": gray.astype(np.float32)}


def dilate_bool_mask(m: np.ndarray, k: int) -> np.ndarray:
    """:583-590"""
    k = _ensure_odd(int(k))
    if k <= 1 or not np.any(m):
        return m
    return cv.dilate(m.astype(np.uint8) * 255, cv.ellipse_se(k), 1) > 127


def color_support_mask(planes: Dict[str, np.ndarray], light_mask: np.ndarray, roi_eff: np.ndarray, sat: np.ndarray, cfg: TempSegConfig = TempSegConfig()):
    """main() :790-797 -> (color_support, chroma)"""
    a, b = planes["a"], planes["b"]
    chroma = np.sqrt((a - 128.0) ** 2 + (b - 128.0) ** 2).astype(np.float32)
    light_d = dilate_bool_mask(light_mask, cfg.color_support_dilate)
    return light_d & roi_eff & (~sat) & (chroma >= float(cfg.color_chroma_min)), chroma


# ---------------------------------------------------------------------------------------------
# Map utilities, fusion, final smoothing: Code/temperature_sensor.py:538-640, :705-747.
# PARITY UNPINNED: the reference tree holds no output of these stages (temperature_map_*.npy are in .MISSING_LARGE_BLOBS) and the
# regressors that feed them only exist as pickles; these restatements follow the source text and OpenCV's documented semantics.
# ---------------------------------------------------------------------------------------------
COLOR_T_MIN, COLOR_T_MAX = 20.0, 33.0            # :55-56
COLOR_GUARD_BAND, SWITCH_MARGIN_C = 0.5, 1.0     # :59-60
FINAL_T_MIN, FINAL_T_MAX = 20.0, 75.0            # :63-64
FINAL_SMOOTH_SIGMA_ACROSS, FINAL_SMOOTH_SIGMA_ALONG = 6.0, 1.0      # :95-96


def clamp_map(m: np.ndarray, roi: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """:538-543"""
    out = m.copy()
    sel = roi & np.isfinite(out)
    out[sel] = np.clip(out[sel], float(lo), float(hi))
    out[~roi] = np.nan
    return out


def inpaint_temperature_map(temp_map: np.ndarray, roi_mask: np.ndarray, radius: int = 7) -> np.ndarray:
    """:546-580: the map is scaled to 8 bits over the range of its known values, cv2.inpaint (Telea) fills the missing pixels of the ROI, and
    EVERY pixel of the ROI is read back from the 8-bit image (the known ones are quantised to 1/255 of the range, as upstream)."""
    out = temp_map.copy()
    inside = roi_mask
    known = inside & np.isfinite(out)
    missing = inside & (~np.isfinite(out))
    if not np.any(missing) or not np.any(known):
        out[~inside] = np.nan
        return out
    vals = out[known]
    vmin = float(np.nanmin(vals))
    vmax = float(np.nanmax(vals))
    if vmax - vmin < 1e-6:
        out[missing] = vmin
        out[~inside] = np.nan
        return out
    scaled = np.zeros_like(out, dtype=np.uint8)
    scaled[known] = ((out[known] - vmin) / (vmax - vmin) * 255.0).clip(0, 255).astype(np.uint8)
    filled = cv.inpaint_telea_u8(scaled, missing.astype(np.uint8) * 255, float(int(radius)))
    out_filled = out.copy()
    out_filled[inside] = (filled[inside].astype(np.float32) / 255.0) * (vmax - vmin) + vmin
    out_filled[~inside] = np.nan
    return out_filled


def fuse_maps_per_pixel(roi: np.ndarray, wide_map: np.ndarray, color_map: np.ndarray):
    """:594-636"""
    final = wide_map.copy()
    source = np.zeros(final.shape, dtype=np.uint8)
    wide_ok = roi & np.isfinite(wide_map)
    with np.errstate(invalid="ignore"):
        color_ok = roi & np.isfinite(color_map) & (color_map >= (COLOR_T_MIN - COLOR_GUARD_BAND)) & (color_map <= (COLOR_T_MAX + COLOR_GUARD_BAND))
        final[color_ok] = color_map[color_ok]
        source[color_ok] = 255
        low_th = COLOR_T_MAX - SWITCH_MARGIN_C
        high_th = COLOR_T_MAX + SWITCH_MARGIN_C
        blend_zone = wide_ok & color_ok & (wide_map > low_th) & (wide_map < high_th)
    if np.any(blend_zone):
        w = (high_th - wide_map[blend_zone]) / (high_th - low_th)
        w = np.clip(w, 0.0, 1.0).astype(np.float32)
        final[blend_zone] = w * color_map[blend_zone] + (1.0 - w) * wide_map[blend_zone]
        source[blend_zone] = 128
    final = clamp_map(final, roi, FINAL_T_MIN, FINAL_T_MAX)
    dbg = {"roi_pixels": int(np.count_nonzero(roi)), "wide_ok_pixels": int(np.count_nonzero(wide_ok)),
           "color_ok_pixels": int(np.count_nonzero(color_ok)), "blend_pixels": int(np.count_nonzero(blend_zone))}
    return final.astype(np.float32), source, dbg


def rotation_matrix_2d(center, angle_deg: float, scale: float = 1.0) -> np.ndarray:
    """cv::getRotationMatrix2D (double)."""
    a = math.radians(float(angle_deg))
    al, be = scale * math.cos(a), scale * math.sin(a)
    cx, cy = float(center[0]), float(center[1])
    return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], np.float64)


def oriented_gaussian_blur_float(map_f: np.ndarray, roi: np.ndarray, angle_rad: float, sigma_across: float, sigma_along: float) -> np.ndarray:
    """:705-747: rotate so that "across the stripes" is +x (warpAffine INTER_LINEAR, BORDER_REFLECT; the ROI with INTER_NEAREST, constant 0),
    anisotropic GaussianBlur, rotate back, NaN outside the twice-rotated ROI."""
    from . import align_oracle as A
    if sigma_across <= 0 and sigma_along <= 0:
        out = map_f.copy()
        out[~roi] = np.nan
        return out
    h, w = map_f.shape
    center = (w / 2.0, h / 2.0)
    angle_deg = -float(angle_rad) * 180.0 / float(np.pi)
    map0 = map_f.copy()
    map0[~np.isfinite(map0)] = 0.0
    roi_u8 = roi.astype(np.uint8) * 255
    M = rotation_matrix_2d(center, angle_deg)
    rot_map = A.warp_affine(map0.astype(np.float32), M, False, border="reflect")
    rot_roi = A.warp_affine(roi_u8, M, False, nearest=True) > 127
    sx, sy = float(max(0.0, sigma_across)), float(max(0.0, sigma_along))
    blurred = cv.gaussian_blur_xy(rot_map.astype(np.float32), sx, sy)
    M_inv = rotation_matrix_2d(center, -angle_deg)
    back = A.warp_affine(blurred, M_inv, False, border="reflect")
    back_roi = A.warp_affine(rot_roi.astype(np.uint8) * 255, M_inv, False, nearest=True) > 127
    out = back.astype(np.float32)
    out[~back_roi] = np.nan
    return out
