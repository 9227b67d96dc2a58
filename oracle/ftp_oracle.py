"""CPU oracle for the VISTAF image -> height-map -> force path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (vistaf-..._amd) never does and fails loudly without its HIP library.

This is a NumPy restatement of /root/reference/Code/shape_ftp.py `main` steps 8-24 (the FTP core
after image loading / alignment) and of the force tail in /root/reference/Code/force_sensor.py.
NumPy calls (fft2, percentile, median, lstsq) are the same calls the reference makes, so dtype
promotion follows NumPy exactly as it would for the reference under the NumPy installed here
(2.2.6) -- with ONE stated exception, the precision of the two FFTs (see FFT_COMPLEX128 below: the
reference pins no NumPy version and the two NumPy generations transform a float32 array in different
precisions).  OpenCV calls are replaced by oracle/cvlite.c (OpenCV is absent and its version unpinned
upstream: "parity unpinned" at the single-stage level, tolerance-checked end to end against the
stored height_map_bundle.npz goldens).

Pinned against (see tests/test_oracle_golden.py, tests/golden/make_golden.py):
  * the reference's own pure-NumPy functions imported in the build container
    (unwrap_quality_guided, robust_polyfit2d, find_top_peaks, choose_carrier_peak,
    refine_peak_parabolic_log, create_circular_mask/apodization, model_predict,
    depth_map_to_volume_cm3, predict_force_from_volume ...) -> tests/golden/ref_numpy_*.npz
  * the 5 stored result.json tails (volume / area / max depth / force)
  * the 5 stored height_map_bundle.npz end-to-end outputs (tolerance, via oracle/align_oracle.py)

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import dataclasses
import json
import math
from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import cvlite as cv

# Precision of the two FFTs of ftp_complex_demod (shape_ftp.py:867, :953).  The reference calls np.fft.fft2 on a float32
# array and pins no NumPy version (README.md:63): NumPy < 2.0 computes that transform in complex128, NumPy >= 2.0 in
# complex64, so the reference's own demodulated field differs between the two by complex64 rounding noise (~1e-7 of the
# spectrum's scale) -- enough to move pixels that sit on a hard threshold (quality >= p25, :749-753).  The oracle
# therefore takes the transform both generations approximate: complex128 (True, the default; exactly what the
# reference computes under NumPy 1.x).  False reproduces NumPy >= 2 (complex64) and is kept for the CPU test that
# measures how far the two are apart (tests/test_oracle_golden.py::test_fft_precision_modes_agree).  The GPU path
# accumulates its pruned DFT in float64 with float64 twiddles, i.e. it follows the complex128 form.
FFT_COMPLEX128 = True


# ---------------------------------------------------------------------------------------------
# Constants: shape_ftp.py:23-218 (as shipped) and force_sensor.py:33-34
# ---------------------------------------------------------------------------------------------
@dataclasses.dataclass
class OracleConfig:
    patch_half_width_bins: int = 10          # :27
    dc_exclusion: int = 10                   # :32
    fft_pad_px: int = 96                     # :35
    pre_blur_sigma_px: float = 1.5           # :38
    roi_erode_px: int = 0                    # :86
    apod_taper_px: int = 120                 # :88
    amp_valid_percentile: float = 25.0       # :90
    quality_smooth_sigma_px: float = 6.0     # :91
    reliable_edge_margin_px: int = 6         # :93
    poly_order: int = 2                      # :95
    reliable_smooth_sigma_px: float = 2.5    # :96
    frontier_zero_band_px: int = 200         # :103
    illum_sigma_px: float = 45.0             # :110
    valid_close_kernel: int = 7              # :114
    valid_close_iters: int = 1               # :115
    bad_pixel_enable: bool = True            # :118
    bad_intensity_percentile: float = 99.9   # :119
    bad_gradient_percentile: float = 99.7    # :120
    bad_dilate_ksize: int = 5                # :121
    bad_dilate_iters: int = 1                # :122
    bad_inpaint_radius: int = 3              # :123
    contact_core_percentile: float = 8.0     # :127
    contact_percentile: float = 92.0         # :128
    dilate_kernel_size: int = 15             # :129
    dilate_iters: int = 2                    # :130
    min_contact_frac: float = 0.002          # :131
    max_contact_frac: float = 0.40           # :132
    hole_neighborhood_px: int = 11           # :140
    hole_known_fraction: float = 0.70        # :141
    hole_min_dist_px: int = 4                # :142
    inpaint_radius: int = 5                  # :144
    unreliable_smooth_sigma_px: float = 9.0  # :148
    contact_blob_min_peak_mm: float = 0.1    # :62
    contact_blob_min_peak_rel_frac: float = 1.0 / 3.0  # :63
    n_fft_peaks: int = 12                    # :168
    peak_max_dy_from_center: float = 0.12    # :203
    plane_order_for_removal: int = 1         # :212
    irls_iters: int = 6                      # :1100
    irls_c: float = 4.685                    # :1100
    grating_pitch_mm: float = 2.0            # force_sensor.py:33
    depth_eps_mm: float = 0.01               # force_sensor.py:34


def cfg_get(cfg: Any, name: str):
    return getattr(cfg, name)


# ---------------------------------------------------------------------------------------------
# ROI helpers: shape_ftp.py:383-414
# ---------------------------------------------------------------------------------------------
def circle_from_3_points(p1, p2, p3) -> Tuple[int, int, int]:
    """shape_ftp.py:406-414"""
    (x1, y1), (x2, y2), (x3, y3) = p1, p2, p3
    a = np.array([[2 * (x2 - x1), 2 * (y2 - y1)], [2 * (x3 - x1), 2 * (y3 - y1)]], dtype=float)
    b = np.array([x2 * x2 + y2 * y2 - x1 * x1 - y1 * y1, x3 * x3 + y3 * y3 - x1 * x1 - y1 * y1], dtype=float)
    cx, cy = np.linalg.solve(a, b)
    r = float(np.hypot(cx - x1, cy - y1))
    return int(round(cx)), int(round(cy)), int(round(r))


def circular_mask(h: int, w: int, cx: int, cy: int, r: int) -> np.ndarray:
    """shape_ftp.py:383-386"""
    yy, xx = np.ogrid[:h, :w]
    return (xx - cx) ** 2 + (yy - cy) ** 2 <= r ** 2


def circular_apodization(h: int, w: int, cx: int, cy: int, r: int, taper_px) -> np.ndarray:
    """shape_ftp.py:389-403 (raised-cosine taper of width taper_px inside radius r)"""
    yy, xx = np.mgrid[0:h, 0:w]
    d = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
    apo = np.zeros((h, w), np.float32)
    r_in = max(0.0, float(r - taper_px))
    flat = d <= r_in
    taper = (d > r_in) & (d <= r)
    apo[flat] = 1.0
    if taper_px > 0:
        t = (d[taper] - r_in) / max(1e-6, float(taper_px))
        apo[taper] = 0.5 * (1.0 + np.cos(np.pi * t))
    return apo


# ---------------------------------------------------------------------------------------------
# small statistics helpers: shape_ftp.py:334-354, :618-622
# ---------------------------------------------------------------------------------------------
def _finite_vals(arr, mask=None):
    v = np.asarray(arr).ravel() if mask is None else np.asarray(arr)[mask]
    return v[np.isfinite(v)]


def nanpercentile_safe(arr, q, mask=None, fallback=None):
    v = _finite_vals(arr, mask)
    if v.size == 0:
        return fallback
    return float(np.nanpercentile(v, q))


def nanmedian_safe(arr, mask=None, fallback=None):
    v = _finite_vals(arr, mask)
    if v.size == 0:
        return fallback
    return float(np.nanmedian(v))


def safe_percentile(vals, q, fallback):
    vals = vals[np.isfinite(vals)]
    if vals.size == 0:
        return fallback
    return float(np.percentile(vals, q))


# ---------------------------------------------------------------------------------------------
# FFT peak search: shape_ftp.py:420-503
# ---------------------------------------------------------------------------------------------
def find_top_peaks(mag, dc_exclusion, n_peaks=10):
    """shape_ftp.py:420-441"""
    mag = np.asarray(mag)
    h, w = mag.shape
    cy, cx = h // 2, w // 2
    work = mag.copy()
    work[max(0, cy - dc_exclusion):min(h, cy + dc_exclusion), max(0, cx - dc_exclusion):min(w, cx + dc_exclusion)] = 0
    flat = work.ravel()
    n_peaks = min(n_peaks, flat.size)
    idx = np.argpartition(flat, -n_peaks)[-n_peaks:]
    idx = idx[np.argsort(flat[idx])[::-1]]
    out = []
    for i in idx:
        y, x = np.unravel_index(i, work.shape)
        out.append((int(x), int(y), float(work[y, x])))
    return out


def choose_carrier_peak(peaks, h, w, peak_max_dy_from_center=0.12):
    """shape_ftp.py:444-463 (FORCE_RIGHT_HALF_PLANE and PREFER_PEAK_NEAR_CENTER_ROW both True)"""
    cy, cx = h // 2, w // 2
    cand = list(peaks)
    right = [p for p in cand if p[0] > cx]
    if right:
        cand = right
    max_dy = int(peak_max_dy_from_center * h)
    near = [p for p in cand if abs(p[1] - cy) <= max_dy]
    if near:
        cand = near
    if not cand:
        cand = list(peaks)
    best = max(cand, key=lambda t: t[2])
    return best[0], best[1]


def _parabolic_subpixel_1d(fm1, f0, fp1):
    den = fm1 - 2.0 * f0 + fp1
    if abs(den) < 1e-12:
        return 0.0
    return 0.5 * (fm1 - fp1) / den


def refine_peak_parabolic_log(mag, peak_x, peak_y):
    """shape_ftp.py:473-483"""
    h, w = mag.shape
    x, y = int(peak_x), int(peak_y)
    if x <= 0 or x >= w - 1 or y <= 0 or y >= h - 1:
        return float(x), float(y)
    lm = np.log(mag + 1e-12)
    dx = _parabolic_subpixel_1d(lm[y, x - 1], lm[y, x], lm[y, x + 1])
    dy = _parabolic_subpixel_1d(lm[y - 1, x], lm[y, x], lm[y + 1, x])
    return float(x + dx), float(y + dy)


# ---------------------------------------------------------------------------------------------
# bad-pixel / glare preprocessing: shape_ftp.py:625-666
# ---------------------------------------------------------------------------------------------
def detect_bad_pixels(gray_f32, valid_mask, cfg):
    """shape_ftp.py:625-649"""
    img = gray_f32.astype(np.float32)
    v = img[valid_mask]
    hi_thr = safe_percentile(v, cfg.bad_intensity_percentile, fallback=np.max(v) if v.size else 255.0)
    gx, gy = cv.sobel3(img)
    grad = np.sqrt(gx * gx + gy * gy)
    g_thr = safe_percentile(grad[valid_mask], cfg.bad_gradient_percentile, fallback=np.max(grad) if v.size else 0.0)
    bad = (img >= hi_thr) | (grad >= g_thr)
    bad &= valid_mask
    if cfg.bad_dilate_ksize and cfg.bad_dilate_ksize > 1:
        ksz = max(3, int(cfg.bad_dilate_ksize) | 1)
        se = cv.ellipse_se(ksz)
        bad = cv.dilate(bad.astype(np.uint8) * 255, se, int(cfg.bad_dilate_iters)) > 0
    return bad, hi_thr, g_thr


def inpaint_float32(img_f32, mask_bool, radius):
    """shape_ftp.py:652-666 (method 'telea'; inputs here are always finite)"""
    if not np.any(mask_bool):
        return img_f32
    return cv.inpaint_telea(img_f32.astype(np.float32), mask_bool.astype(np.uint8) * 255, float(radius))


# ---------------------------------------------------------------------------------------------
# ftp_complex_demod: shape_ftp.py:810-1037  (FFT_SIDEBAND_METHOD == "patch_shift")
# ---------------------------------------------------------------------------------------------
def preprocess_for_fft(gray_crop_u8, apo, cfg):
    """shape_ftp.py:821-862: bad-pixel inpaint, illumination normalise, pre-blur, apodise,
    median removal, reflect pad.  Returns (Iw_fft, intermediates)."""
    inter = {}
    img0 = np.asarray(gray_crop_u8).astype(np.float32)
    if cfg.bad_pixel_enable:
        valid = apo > 1e-6
        bad, hi_thr, g_thr = detect_bad_pixels(img0, valid, cfg)
        inter["bad"] = bad
        inter["hi_thr"] = hi_thr
        inter["g_thr"] = g_thr
        if np.any(bad):
            img0 = inpaint_float32(img0, bad, cfg.bad_inpaint_radius)
    inter["img_inpainted"] = img0
    blur = cv.gaussian_blur(img0, cfg.illum_sigma_px)
    i_norm = img0 / (blur + 1e-6) - 1.0
    if cfg.pre_blur_sigma_px and cfg.pre_blur_sigma_px > 0:
        i_norm = cv.gaussian_blur(i_norm, cfg.pre_blur_sigma_px).astype(np.float32)
    iw = i_norm * apo
    mu = nanmedian_safe(iw, mask=(apo > 1e-6), fallback=0.0)
    iw = iw - mu
    inter["mu"] = mu
    inter["iw"] = iw
    pad = int(max(0, cfg.fft_pad_px))
    iw_fft = cv.pad_reflect(iw, pad) if pad > 0 else iw
    return iw_fft, inter


def hann_patch_window(hp, wp):
    """shape_ftp.py:800-807"""
    wy = np.hanning(hp).astype(np.float32)
    wx = np.hanning(wp).astype(np.float32)
    return (wy[:, None] * wx[None, :]).astype(np.float32)


def ftp_complex_demod(gray_crop_u8, apo, cfg, locked_peak_refined=None):
    """shape_ftp.py:810-1037.  locked_peak_refined=None -> carrier searched (reference frame, :878-883);
    otherwise carrier LOCKED to the given refined peak (:891-894)."""
    iw_fft, inter = preprocess_for_fft(gray_crop_u8, apo, cfg)
    h, w = np.asarray(gray_crop_u8).shape
    pad = int(max(0, cfg.fft_pad_px))
    hf, wf = iw_fft.shape
    cy, cx = hf // 2, wf // 2

    F = np.fft.fft2(iw_fft.astype(np.float64) if FFT_COMPLEX128 else iw_fft)
    F_shift = np.fft.fftshift(F)
    fft_mag = np.abs(F_shift)

    if locked_peak_refined is None:
        peaks = find_top_peaks(fft_mag, dc_exclusion=cfg.dc_exclusion, n_peaks=cfg.n_fft_peaks)
        peak_x, peak_y = choose_carrier_peak(peaks, hf, wf, cfg.peak_max_dy_from_center)
        peak_x_f, peak_y_f = refine_peak_parabolic_log(fft_mag, peak_x, peak_y)
    else:
        peak_x_f, peak_y_f = float(locked_peak_refined[0]), float(locked_peak_refined[1])

    kx = peak_x_f - cx
    ky = peak_y_f - cy

    px_i = int(np.round(peak_x_f))
    py_i = int(np.round(peak_y_f))
    bw = int(max(3, cfg.patch_half_width_bins))
    x0, x1 = max(0, px_i - bw), min(wf, px_i + bw + 1)
    y0, y1 = max(0, py_i - bw), min(hf, py_i + bw + 1)
    patch = F_shift[y0:y1, x0:x1].copy()
    ph, pw = patch.shape
    patch *= hann_patch_window(ph, pw)
    F_demod_shift = np.zeros_like(F_shift)
    cy0 = cy - ph // 2
    cx0 = cx - pw // 2
    F_demod_shift[cy0:cy0 + ph, cx0:cx0 + pw] = patch
    complex_field = np.fft.ifft2(np.fft.ifftshift(F_demod_shift))

    dpx = float(peak_x_f - px_i)
    dpy = float(peak_y_f - py_i)
    if abs(dpx) > 1e-6 or abs(dpy) > 1e-6:
        yy, xx = np.mgrid[0:hf, 0:wf]
        complex_field = complex_field * np.exp(-1j * 2.0 * np.pi * (dpx * (xx / wf) + dpy * (yy / hf)))

    cdemod = complex_field[pad:pad + h, pad:pad + w] if pad > 0 else complex_field
    amp = np.abs(cdemod).astype(np.float32)
    return {
        "field": cdemod,
        "amp": amp,
        "peak_refined": (float(peak_x_f), float(peak_y_f)),
        "peak_int": (int(round(peak_x_f)), int(round(peak_y_f))),
        "k": (float(kx), float(ky)),
        "fft_shape": (hf, wf),
        "patch": patch,
        "inter": inter,
    }


# ---------------------------------------------------------------------------------------------
# reliable mask: shape_ftp.py:707-775
# ---------------------------------------------------------------------------------------------
def largest_connected_component(mask_bool):
    """shape_ftp.py:707-718"""
    if np.count_nonzero(mask_bool) == 0:
        return mask_bool
    num, labels, areas = cv.cc8(mask_bool)
    if num <= 1:
        return mask_bool
    best = 1 + int(np.argmax(areas[1:]))
    return labels == best


def erode_by_distance(mask_bool, margin_px):
    """shape_ftp.py:721-726"""
    if margin_px <= 0:
        return mask_bool
    dist = cv.dist_l2_3x3(mask_bool.astype(np.uint8) * 255)
    return (dist > float(margin_px)) & mask_bool


def compute_reliable_mask(amp_ref, amp_def, roi_eroded, circ_mask, cfg):
    """shape_ftp.py:739-775"""
    amp_prod = (amp_ref * amp_def).astype(np.float32)
    quality = amp_prod
    if cfg.quality_smooth_sigma_px and cfg.quality_smooth_sigma_px > 0:
        quality = cv.gaussian_blur(quality, cfg.quality_smooth_sigma_px).astype(np.float32)
    amp_thr = nanpercentile_safe(quality, cfg.amp_valid_percentile, mask=roi_eroded, fallback=None)
    if amp_thr is None:
        amp_thr = nanpercentile_safe(quality, cfg.amp_valid_percentile, mask=circ_mask, fallback=0.0)
    reliable = roi_eroded & (quality >= float(amp_thr)) & np.isfinite(quality)
    thresholded = reliable.copy()
    if np.any(reliable):
        ksz = max(3, int(cfg.valid_close_kernel) | 1)
        se = cv.ellipse_se(ksz)
        closed = cv.morph_close(reliable.astype(np.uint8) * 255, se, int(cfg.valid_close_iters))
        reliable = (closed > 0) & roi_eroded
    if np.any(reliable):
        reliable = largest_connected_component(reliable) & roi_eroded
    if cfg.reliable_edge_margin_px and cfg.reliable_edge_margin_px > 0 and np.any(reliable):
        reliable = erode_by_distance(reliable, cfg.reliable_edge_margin_px)
    return reliable, quality, float(amp_thr), thresholded


# ---------------------------------------------------------------------------------------------
# unwrap: shape_ftp.py:1043-1080
# ---------------------------------------------------------------------------------------------
def unwrap_quality_guided(wrapped, mask, quality):
    """C restatement with the exact heapq tuple order (oracle/cvlite.c)."""
    return cv.unwrap_quality_guided(wrapped, mask, quality)


def unwrap_quality_guided_py(wrapped, mask, quality):
    """Literal pure-Python form of shape_ftp.py:1043-1080, for small cross-checks only."""
    import heapq

    h, w = wrapped.shape
    out = np.full((h, w), np.nan, np.float32)
    m = mask.astype(bool)
    if not np.any(m):
        return out
    q = quality.copy().astype(np.float32)
    q[~m] = -np.inf
    sy, sx = np.unravel_index(np.argmax(q), q.shape)
    out[sy, sx] = wrapped[sy, sx]
    heap = []
    nbrs = [(-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (-1, 1), (1, -1), (1, 1)]

    def push(py, px):
        for dy, dx in nbrs:
            ny, nx = py + dy, px + dx
            if 0 <= ny < h and 0 <= nx < w and m[ny, nx] and not np.isfinite(out[ny, nx]):
                heapq.heappush(heap, (-float(q[ny, nx]), ny, nx, py, px))

    push(sy, sx)
    while heap:
        _, y, x, py, px = heapq.heappop(heap)
        if not m[y, x] or np.isfinite(out[y, x]) or not np.isfinite(out[py, px]):
            continue
        dw = np.angle(np.exp(1j * (wrapped[y, x] - wrapped[py, px])))
        out[y, x] = out[py, px] + dw
        push(y, x)
    return out


# ---------------------------------------------------------------------------------------------
# robust polynomial fit: shape_ftp.py:1086-1136
# ---------------------------------------------------------------------------------------------
def _design(xn, yn, order):
    cols = [xn, yn, np.ones_like(xn)]
    if order >= 2:
        cols += [xn * xn, xn * yn, yn * yn]
    return np.stack(cols, axis=1)


def _eval_poly(xn, yn, coef, order):
    z = coef[0] * xn + coef[1] * yn + coef[2]
    if order >= 2:
        z = z + coef[3] * xn * xn + coef[4] * xn * yn + coef[5] * yn * yn
    return z


def robust_polyfit2d(z, mask, order=2, iters=6, c=4.685):
    """shape_ftp.py:1100-1136: IRLS with weights 1/(1+u^2), sigma = 1.4826*MAD."""
    h, w = z.shape
    m = mask & np.isfinite(z)
    if np.count_nonzero(m) < 200:
        ncoef = 6 if order >= 2 else 3
        return np.zeros((ncoef,), np.float32), np.zeros_like(z, np.float32)
    yy, xx = np.indices((h, w))
    x = xx[m].astype(np.float32)
    y = yy[m].astype(np.float32)
    zz = z[m].astype(np.float32)
    xn = (x - (w - 1) / 2.0) / ((w - 1) / 2.0)
    yn = (y - (h - 1) / 2.0) / ((h - 1) / 2.0)
    a = _design(xn, yn, order)
    wts = np.ones_like(zz, np.float32)
    for _ in range(iters):
        coef, *_ = np.linalg.lstsq(a * wts[:, None], zz * wts, rcond=None)
        r = zz - (a @ coef)
        med = np.median(r)
        mad = np.median(np.abs(r - med)) + 1e-6
        sigma = 1.4826 * mad
        u = r / (c * sigma)
        wts = 1.0 / (1.0 + u * u)
    coef = coef.astype(np.float32)
    yyf, xxf = np.indices((h, w))
    xnf = (xxf.astype(np.float32) - (w - 1) / 2.0) / ((w - 1) / 2.0)
    ynf = (yyf.astype(np.float32) - (h - 1) / 2.0) / ((h - 1) / 2.0)
    fit = _eval_poly(xnf, ynf, coef, order).astype(np.float32)
    return coef, fit


# ---------------------------------------------------------------------------------------------
# smoothing, frontier taper, clamp, calibration curve, blob filter
# ---------------------------------------------------------------------------------------------
def masked_gaussian_smooth(z, mask, sigma):
    """shape_ftp.py:1139-1147 (normalised convolution)"""
    if sigma <= 0:
        return z
    z0 = z.copy().astype(np.float32)
    m = mask.astype(np.float32)
    z0[~mask] = 0.0
    num = cv.gaussian_blur(z0, sigma)
    den = cv.gaussian_blur(m, sigma) + 1e-6
    return (num / den).astype(np.float32)


def compute_internal_holes_within_mask(container_mask, known_mask, ksize, frac_thr, min_dist_edge_px):
    """shape_ftp.py:1153-1175"""
    container = container_mask.astype(bool)
    known = known_mask.astype(bool) & container
    holes = container & (~known)
    if not np.any(holes):
        return np.zeros_like(container, dtype=bool)
    k = max(3, int(ksize) | 1)
    count_known = cv.box_sum(known.astype(np.float32), k)
    count_cont = cv.box_sum(container.astype(np.float32), k)
    frac = count_known / (count_cont + 1e-6)
    dist = cv.dist_l2_3x3(container.astype(np.uint8) * 255)
    return holes & (frac >= float(frac_thr)) & (dist >= float(min_dist_edge_px))


def inpaint_only_mask(z_known, roi_mask, inpaint_mask, radius):
    """shape_ftp.py:1178-1203"""
    z = z_known.astype(np.float32)
    roi = roi_mask.astype(bool)
    m = inpaint_mask.astype(bool) & roi
    out = z.copy()
    out[~roi] = np.nan
    if not np.any(m):
        return out
    known = roi & (~m) & np.isfinite(z)
    fill_val = float(np.nanmedian(z[known])) if np.any(known) else 0.0
    zin = np.full_like(z, fill_val, dtype=np.float32)
    zin[known] = z[known]
    zout = cv.inpaint_telea(zin, m.astype(np.uint8) * 255, float(radius)).astype(np.float32)
    out[m] = zout[m]
    out[~roi] = np.nan
    return out


def _smoothstep01(t):
    """shape_ftp.py:1277-1284 (FRONTIER_ZERO_CURVE == 'smoothstep')"""
    t = np.clip(t, 0.0, 1.0).astype(np.float32)
    return (t * t * (3.0 - 2.0 * t)).astype(np.float32)


def apply_frontier_zero_transition(height, roi_mask, reliable_mask, band_px, base_value=0.0,
                                   apply_inside=True, apply_outside=True):
    """shape_ftp.py:1287-1324"""
    out = height.astype(np.float32).copy()
    roi = roi_mask.astype(bool)
    rel = reliable_mask.astype(bool) & roi
    if (not np.any(rel)) or (band_px is None) or (float(band_px) <= 0):
        return out
    band = float(band_px)
    dist_in = cv.dist_l2_3x3(rel.astype(np.uint8) * 255).astype(np.float32)
    dist_in_edge = np.maximum(dist_in - 1.0, 0.0)
    dist_out = cv.dist_l2_3x3((~rel).astype(np.uint8) * 255).astype(np.float32)
    dist_out_edge = np.maximum(dist_out - 1.0, 0.0)
    if apply_inside:
        inside = rel & np.isfinite(out)
        wgt = _smoothstep01(dist_in_edge / max(1e-6, band))
        out[inside] = float(base_value) + (out[inside] - float(base_value)) * wgt[inside]
    if apply_outside:
        outside_band = roi & (~rel) & (dist_out_edge <= band)
        out[outside_band] = float(base_value)
    return out


def clamp_positive_to_zero(z, mask):
    """shape_ftp.py:1206-1213"""
    out = z.astype(np.float32).copy()
    m = mask.astype(bool) & np.isfinite(out)
    out[m] = np.minimum(out[m], 0.0)
    return out


def load_calibration(json_path: str):
    """shape_ftp.py:672-680"""
    with open(json_path, "r", encoding="utf-8") as f:
        cal = json.load(f)
    return cal["best_model"], bool(cal.get("use_negated_height_for_fit", True))


def model_predict(model, xs):
    """shape_ftp.py:682-700"""
    xs = np.asarray(xs, float)
    t = model["type"]
    p = model["params"]
    xs = np.maximum(xs, 0.0)
    if t == "growth":
        a = float(p["a"]); b = float(p["b"])
        return a * (np.exp(b * xs) - 1.0)
    if t == "hinge_saturating":
        a = float(p["a"]); b = float(p["b"]); c = float(p["c"])
        return a * ((1.0 - np.exp(-b * np.maximum(xs - c, 0.0))) - (1.0 - np.exp(-b * np.maximum(0.0 - c, 0.0))))
    raise ValueError(f"Unknown model type in calibration: {t}")


def height_unitless_to_depth_mm(height_unitless, model, use_negated_height=True):
    """shape_ftp.py:702-705"""
    h = np.asarray(height_unitless, dtype=np.float32)
    x = (-h) if use_negated_height else h
    return model_predict(model, x).astype(np.float32)


def filter_blobs_by_peak_depth_mm(height_mm, roi_mask, min_peak_mm, min_peak_rel_frac):
    """shape_ftp.py:1215-1271 (MM_KEEP_INDENTATION_NEGATIVE False, min_area 0, removed -> zero)"""
    out = height_mm.astype(np.float32).copy()
    roi = roi_mask.astype(bool) & np.isfinite(out)
    depth = out.astype(np.float32)
    cand = roi & (depth > 0.0)
    if not np.any(cand):
        return out, np.zeros_like(roi_mask, dtype=bool)
    global_max_peak = float(np.max(depth[cand]))
    thr = float(min_peak_mm)
    if (min_peak_rel_frac is not None) and np.isfinite(global_max_peak):
        thr = max(thr, float(min_peak_rel_frac) * global_max_peak)
    num, labels, _ = cv.cc8(cand)
    peaks = np.full(num, -np.inf, np.float32)
    np.maximum.at(peaks, labels[cand], depth[cand])
    keep = peaks >= thr
    keep[0] = False
    kept_mask = keep[labels] & cand
    removed = cand & (~kept_mask)
    out[removed] = 0.0
    return out, kept_mask


# ---------------------------------------------------------------------------------------------
# force tail: force_sensor.py:93-187
# ---------------------------------------------------------------------------------------------
def depth_map_to_volume_cm3(height_map_mm, roi_mask, mm_per_px, depth_eps_mm=0.01):
    """force_sensor.py:93-123"""
    z = np.asarray(height_map_mm, dtype=np.float32).copy()
    roi = np.asarray(roi_mask, dtype=bool)
    pos = np.clip(z, 0.0, np.inf)
    neg = np.clip(-z, 0.0, np.inf)
    depth = neg if float(np.nansum(neg)) > float(np.nansum(pos)) else pos
    depth[~roi] = 0.0
    depth = np.where(np.isfinite(depth), depth, 0.0).astype(np.float32)
    contact = depth > float(depth_eps_mm)
    if not np.any(contact):
        return 0.0, 0.0, 0.0
    pixel_area_mm2 = float(mm_per_px) ** 2
    volume_mm3 = float(np.sum(depth[contact]) * pixel_area_mm2)
    area_mm2 = float(np.count_nonzero(contact) * pixel_area_mm2)
    max_depth_mm = float(np.max(depth[contact]))
    return float(volume_mm3 / 1000.0), float(area_mm2), float(max_depth_mm)


def predict_force_from_volume(best_model: Dict[str, Any], volume_cm3: float) -> float:
    """force_sensor.py:129-167"""
    t = best_model["type"]
    p = best_model["params"]
    v = float(volume_cm3)
    if t == "linear0":
        return float(float(p["a"]) * v)
    if t == "linear":
        return float(float(p["a"]) * v + float(p["b"]))
    if t == "poly2":
        return float(float(p["c2"]) * v * v + float(p["c1"]) * v + float(p["c0"]))
    if t == "sat_exp":
        return float(float(p["a"]) * (1.0 - np.exp(-float(p["b"]) * np.maximum(v, 0.0))))
    if t == "growth":
        return float(float(p["a"]) * (np.exp(float(p["b"]) * np.maximum(v, 0.0)) - 1.0))
    if t == "hinge_saturating":
        a, b, c = float(p["a"]), float(p["b"]), float(p["c"])
        vv = np.asarray(v, float)
        return float(a * ((1.0 - np.exp(-b * np.maximum(vv - c, 0.0))) - (1.0 - np.exp(-b * np.maximum(0.0 - c, 0.0)))))
    raise ValueError(f"Unknown model type in force calibration JSON: {t}")


def estimate_mm_per_px(period_px: Optional[float], grating_pitch_mm: float = 2.0) -> float:
    """force_sensor.py:173-187"""
    if period_px is None:
        raise RuntimeError("shape_ftp did not return estimated_grating_period_px and OVERRIDE_MM_PER_PX is not set.")
    est = float(period_px)
    if (not np.isfinite(est)) or est <= 1e-12:
        raise RuntimeError(f"Invalid estimated_grating_period_px={period_px}.")
    return float(grating_pitch_mm) / est


# ---------------------------------------------------------------------------------------------
# arg-extremum ("contact-location index")
# ---------------------------------------------------------------------------------------------
def argmax_depth_mm(height_mm, roi):
    """shape_ftp.py:1945-1959 (MIN_DEPTH_SCOPE 'roi', positive depth): nanargmax, row-major first."""
    m = roi & np.isfinite(height_mm)
    if not np.any(m):
        return -1
    vals = height_mm.copy()
    vals[~m] = np.nan
    return int(np.nanargmax(vals))


def compute_min_height(height_final, mask):
    """phase_to_height.py:1009-1016: (value, (x, y)) of the minimum unitless height."""
    m = mask.astype(bool) & np.isfinite(height_final)
    if not np.any(m):
        return np.nan, None
    tmp = np.full_like(height_final, np.inf, dtype=np.float32)
    tmp[m] = height_final[m].astype(np.float32)
    iy, ix = np.unravel_index(int(np.argmin(tmp)), tmp.shape)
    return float(height_final[iy, ix]), (int(ix), int(iy))


# ---------------------------------------------------------------------------------------------
# reference-frame state + per-frame path  (shape_ftp.main :1497-1528, :1632-2037)
# ---------------------------------------------------------------------------------------------
def make_reference_state(ref_gray_u8, cx, cy, r, cfg) -> Dict[str, Any]:
    """Everything main() derives from the reference frame alone (:1514-1528, :1632-1639)."""
    h, w = ref_gray_u8.shape
    circ = circular_mask(h, w, cx, cy, r)
    r_valid = max(0, r - int(cfg.roi_erode_px))
    roi = circular_mask(h, w, cx, cy, r_valid)
    apo = circular_apodization(h, w, cx, cy, r, cfg.apod_taper_px)
    dem = ftp_complex_demod(ref_gray_u8, apo, cfg, locked_peak_refined=None)
    return {"circ": circ, "roi": roi, "apo": apo, "demod": dem, "shape": (h, w), "circle": (cx, cy, r)}


def process_frame(def_gray_u8, ref_state, cfg, cal_model, cal_use_neg=True,
                  force_model=None, keep_intermediates=False) -> Optional[Dict[str, Any]]:
    """shape_ftp.main :1641-2037 for one (already aligned) deformed crop with LOCK_CARRIER_TO_REFERENCE,
    followed by the force tail as called by multimodal_sensor.py:388-419 (roi = isfinite(height))."""
    h, w = ref_state["shape"]
    roi = ref_state["roi"]
    circ = ref_state["circ"]
    apo = ref_state["apo"]
    rd = ref_state["demod"]
    inter: Dict[str, Any] = {}

    dd = ftp_complex_demod(def_gray_u8, apo, cfg, locked_peak_refined=rd["peak_refined"])
    cref, cdef = rd["field"], dd["field"]
    k_ref, k_def = rd["k"], dd["k"]
    dkx = k_def[0] - k_ref[0]
    dky = k_def[1] - k_ref[1]
    hf, wf = dd["fft_shape"]

    reliable, quality, amp_thr, thresholded = compute_reliable_mask(rd["amp"], dd["amp"], roi, circ, cfg)
    if not np.any(reliable):
        return None  # :1677-1679

    ratio = cdef * np.conj(cref)
    if (abs(dkx) > 1e-6) or (abs(dky) > 1e-6):
        yy, xx = np.mgrid[0:h, 0:w]
        ratio = ratio * np.exp(1j * 2.0 * np.pi * (dkx * (xx / max(1, wf)) + dky * (yy / max(1, hf))))
    wrapped = np.angle(ratio).astype(np.float32)

    unwrapped = unwrap_quality_guided(wrapped, reliable, quality)

    # debug_ramp (:1357-1422) -- functional: subtracts an order-1 robust fit
    phase = unwrapped.copy()
    phase[~reliable] = np.nan
    if int(cfg.plane_order_for_removal) > 0 and phase[reliable].size >= 500:      # 0: no debug_ramp (Code/phase_to_height.py has none)
        _, fit1 = robust_polyfit2d(phase, reliable, order=int(cfg.plane_order_for_removal), iters=cfg.irls_iters, c=cfg.irls_c)
        phase = (phase - fit1).astype(np.float32)
    deramped = phase

    # two-pass detrend (:1716-1751)
    coef0, fit0 = robust_polyfit2d(phase, reliable, order=cfg.poly_order, iters=cfg.irls_iters, c=cfg.irls_c)
    residual0 = (phase - fit0).astype(np.float32)
    abs_res = np.abs(residual0).astype(np.float32)
    thr = nanpercentile_safe(abs_res, cfg.contact_percentile, mask=reliable, fallback=None)
    if thr is None or not np.isfinite(thr):
        thr = nanpercentile_safe(abs_res, 95, mask=reliable, fallback=0.0)
    contact = (abs_res >= float(thr)) & reliable & np.isfinite(abs_res)
    frac = contact.sum() / max(1, reliable.sum())
    if frac < cfg.min_contact_frac:
        thr2 = nanpercentile_safe(abs_res, 95, mask=reliable, fallback=thr)
        contact = (abs_res >= float(thr2)) & reliable & np.isfinite(abs_res)
    elif frac > cfg.max_contact_frac:
        thr2 = nanpercentile_safe(abs_res, 98, mask=reliable, fallback=thr)
        contact = (abs_res >= float(thr2)) & reliable & np.isfinite(abs_res)
    se = cv.ellipse_se(int(cfg.dilate_kernel_size))
    contact_d = (cv.dilate(contact.astype(np.uint8) * 255, se, int(cfg.dilate_iters)) > 0) & reliable
    background = reliable & (~contact_d)
    if background.sum() < int(0.15 * reliable.sum()):
        background = reliable.copy()
    coef, fit = robust_polyfit2d(phase, background, order=cfg.poly_order, iters=cfg.irls_iters, c=cfg.irls_c)
    detrended = (phase - fit).astype(np.float32)
    bg_med = nanmedian_safe(detrended, mask=background, fallback=None)
    if bg_med is None or not np.isfinite(bg_med):
        bg_med = nanmedian_safe(detrended, mask=reliable, fallback=0.0)
    zeroed = detrended - float(bg_med)

    # reliable-only smoothing (:1755-1757)
    height_map = zeroed.copy()
    if cfg.reliable_smooth_sigma_px and cfg.reliable_smooth_sigma_px > 0:
        height_map = masked_gaussian_smooth(height_map, reliable & np.isfinite(height_map), cfg.reliable_smooth_sigma_px)

    # auto sign flip (:1759-1768)
    flipped = False
    core_thr = nanpercentile_safe(height_map, cfg.contact_core_percentile, mask=reliable, fallback=None)
    if core_thr is not None and np.isfinite(core_thr):
        core = reliable & np.isfinite(height_map) & (height_map <= float(core_thr))
        if np.any(core):
            med_core = float(np.median(height_map[core]))
            if med_core > 0:
                height_map *= -1.0
                flipped = True

    # holes (:1770-1799)
    known_height = reliable & np.isfinite(height_map)
    height_rel = np.full((h, w), np.nan, np.float32)
    height_rel[known_height] = height_map[known_height]
    holes = compute_internal_holes_within_mask(reliable, known_height, cfg.hole_neighborhood_px,
                                               cfg.hole_known_fraction, cfg.hole_min_dist_px)
    if np.any(holes):
        tmp = height_rel.copy()
        med = float(np.nanmedian(tmp[known_height])) if np.any(known_height) else 0.0
        tmp[reliable & ~known_height] = med
        filled = inpaint_only_mask(tmp, reliable, holes, cfg.inpaint_radius)
        height_rel[holes] = filled[holes]
    output_reliable = reliable & np.isfinite(height_rel)

    # frontier taper inside reliable (:1803-1814)
    if cfg.frontier_zero_band_px and cfg.frontier_zero_band_px > 0:
        height_rel = apply_frontier_zero_transition(height_rel, roi, output_reliable, cfg.frontier_zero_band_px,
                                                    apply_inside=True, apply_outside=False)

    # compose (:1816-1841)
    height_final = np.full((h, w), np.nan, np.float32)
    height_final[roi] = 0.0
    height_final[output_reliable] = height_rel[output_reliable]
    if cfg.unreliable_smooth_sigma_px and cfg.unreliable_smooth_sigma_px > 0:
        smooth_all = masked_gaussian_smooth(height_final, roi, cfg.unreliable_smooth_sigma_px)
        upd = roi & (~output_reliable)
        height_final[upd] = smooth_all[upd]
    if cfg.frontier_zero_band_px and cfg.frontier_zero_band_px > 0:
        height_final = apply_frontier_zero_transition(height_final, roi, output_reliable, cfg.frontier_zero_band_px,
                                                      apply_inside=False, apply_outside=True)
    height_final = clamp_positive_to_zero(height_final, roi)
    height_unitless = height_final

    # unitless -> mm (:1850-1855) and blob filter (:1862-1873)
    depth_mm = height_unitless_to_depth_mm(height_unitless, cal_model, cal_use_neg)
    height_out, contact_kept = filter_blobs_by_peak_depth_mm(depth_mm, roi, cfg.contact_blob_min_peak_mm,
                                                             cfg.contact_blob_min_peak_rel_frac)

    # period estimate (:2015-2027)
    vals = []
    if abs(float(k_ref[0])) > 1e-9:
        vals.append(float(rd["fft_shape"][1]) / abs(float(k_ref[0])))
    if abs(float(k_def[0])) > 1e-9:
        vals.append(float(dd["fft_shape"][1]) / abs(float(k_def[0])))
    period = float(np.mean(vals)) if vals else None

    out: Dict[str, Any] = {
        "height_map_mm_crop": height_out,
        "roi_eroded_crop": roi,
        "output_reliable_crop": output_reliable,
        "estimated_grating_period_px": period,
        "height_unitless": height_unitless,
        "reliable": reliable,
        "contact_dilated": contact_d,
        "contact_kept_by_depth": contact_kept,
        "hole_candidates": holes,
        "argmax_depth_index": argmax_depth_mm(height_out, roi),
        "argmin_unitless": compute_min_height(height_unitless, roi),
        "flipped": flipped,
    }
    if force_model is not None and period is not None:
        mm_per_px = estimate_mm_per_px(period, cfg.grating_pitch_mm)
        v, a, md = depth_map_to_volume_cm3(height_out, np.isfinite(height_out), mm_per_px, cfg.depth_eps_mm)
        out.update(mm_per_px=mm_per_px, volume_cm3=v, contact_area_mm2=a, max_depth_mm=md,
                   force_N=predict_force_from_volume(force_model, v))
    if keep_intermediates:
        out["inter"] = {
            "demod": dd, "quality": quality, "amp_thr": amp_thr, "thresholded": thresholded,
            "wrapped": wrapped, "unwrapped": unwrapped, "deramped": deramped, "residual0": residual0,
            "contact_thr": thr, "background": background, "coef": coef, "bg_med": bg_med,
            "zeroed": zeroed, "height_smooth": height_map, "core_thr": core_thr, "height_rel": height_rel,
        }
    return out
