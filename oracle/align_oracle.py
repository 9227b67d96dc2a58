"""CPU restatement of the reference's pre-path: image decode, global phase-correlation shift, ECC crop alignment.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): the product never imports this module.

Follows Code/shape_ftp.py:1471-1537 (main: imread, BGR2GRAY, estimate_global_shift, warpAffine, ROI crop,
align_crop_ecc) and the OpenCV 4.x routines those lines call, restated from their published algorithms:

* cv2.cvtColor(BGR2GRAY) on uint8: fixed point.  OpenCV 4.x: (B*3735 + G*19235 + R*9798 + 2^14) >> 15; OpenCV 3.x:
  (R*4899 + G*9617 + B*1868 + 2^13) >> 14 -- they differ by one unit for 0.26 % of all colours, and the reference pins no
  version.  The stored outputs decide per data set: the 75 loading photographs and the five demo pairs (the force path) are
  reproduced better by the 4.x form (median relative volume error 9.9e-5 against 1.7e-4; tests/golden/e2e_loading_report.json),
  the four phase-to-height calibration photographs by the 3.x form (the 2 mm minimum to 2.6e-5 and its pixel exactly, against
  2.5e-4 and one row off).  `generation` selects; 4 is the default and what the path's own colour input uses;
* cv2.createHanningWindow: sqrt(hann_row * hann_col) in float32 (the implementation ends with cv::sqrt);
* cv2.phaseCorrelate: windowed DFTs, unit-magnitude cross-power spectrum, inverse DFT, fftshift, arg-max,
  5x5 intensity-weighted centroid, shift = centre - centroid;
* cv2.warpAffine(INTER_LINEAR): source coordinates in fixed point (AB_BITS = 10) rounded to 1/32 pixel, bilinear
  weights from the 32x32 table; uint8 results through the 15-bit integer table, float32 results in float;
* cv2.findTransformECC(MOTION_EUCLIDEAN, gaussFiltSize = 1): Evangelidis & Psarakis' forward-additive ECC iteration.

Parity status: the decoder is Pillow's libjpeg-turbo instead of OpenCV's copy of it, and float reductions are
NumPy's instead of OpenCV's SIMD loops, so the alignment is a tolerance-level restatement (sub-1e-3 px), not a
bit-exact one.  It exists so that the path oracle (ftp_oracle.py) can be run end to end on the reference's own
demo photographs and compared with the height-map bundles the reference stored for them (tests/test_e2e_bundles.py).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

from . import cvlite

AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def imread_bgr(path: str) -> np.ndarray:
    """cv2.imread(path, IMREAD_COLOR): 8-bit BGR (decoder: Pillow / libjpeg-turbo)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


def bgr2gray_u8(bgr: np.ndarray, generation: int = 4) -> np.ndarray:
    """generation 4: OpenCV 4.x's 15-bit coefficients (default); 3: OpenCV 3.x's 14-bit ones (module docstring)"""
    b = bgr[..., 0].astype(np.int32)
    g = bgr[..., 1].astype(np.int32)
    r = bgr[..., 2].astype(np.int32)
    if int(generation) == 3:
        return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def hanning_window(h: int, w: int) -> np.ndarray:
    wc = 0.5 * (1.0 - np.cos(2.0 * np.pi / (w - 1) * np.arange(w, dtype=np.float64)))
    wr = 0.5 * (1.0 - np.cos(2.0 * np.pi / (h - 1) * np.arange(h, dtype=np.float64)))
    return np.sqrt((wr[:, None] * wc[None, :]).astype(np.float32))


def _optimal_dft_size(n: int) -> int:
    """cv::getOptimalDFTSize: smallest m >= n with m = 2^a 3^b 5^c."""
    m = n
    while True:
        k = m
        for p in (2, 3, 5):
            while k % p == 0:
                k //= p
        if k == 1:
            return m
        m += 1


def phase_correlate(a: np.ndarray, b: np.ndarray, window: Optional[np.ndarray]) -> Tuple[Tuple[float, float], float]:
    h, w = a.shape
    M, N = _optimal_dft_size(h), _optimal_dft_size(w)
    pa = np.zeros((M, N), np.float32)
    pb = np.zeros((M, N), np.float32)
    pa[:h, :w] = a
    pb[:h, :w] = b
    if window is not None:
        pw = np.zeros((M, N), np.float32)
        pw[:h, :w] = window
        pa *= pw
        pb *= pw
    f1 = np.fft.fft2(pa)
    f2 = np.fft.fft2(pb)
    p = f1 * np.conj(f2)
    mag = np.abs(p)
    c = np.where(mag > 0, p / np.maximum(mag, np.finfo(np.float32).tiny), 0)
    corr = np.fft.fftshift(np.real(np.fft.ifft2(c))).astype(np.float32) * np.float32(M * N)   # cv::idft without DFT_SCALE
    py, px = np.unravel_index(int(np.argmax(corr)), corr.shape)
    y0, y1 = max(0, py - 2), min(M - 1, py + 2)
    x0, x1 = max(0, px - 2), min(N - 1, px + 2)
    win = corr[y0:y1 + 1, x0:x1 + 1].astype(np.float64)
    s = float(win.sum())
    xs = np.arange(x0, x1 + 1, dtype=np.float64)
    ys = np.arange(y0, y1 + 1, dtype=np.float64)
    cx = float((win * xs[None, :]).sum()) / s
    cy = float((win * ys[:, None]).sum()) / s
    response = s / float(M * N)
    return (N / 2.0 - cx, M / 2.0 - cy), response


def estimate_global_shift(ref_gray_f32: np.ndarray, def_gray_f32: np.ndarray):
    """shape_ftp.py:529-535."""
    rb = cvlite.gaussian_blur(ref_gray_f32, 7.0)
    db = cvlite.gaussian_blur(def_gray_f32, 7.0)
    h, w = rb.shape
    return phase_correlate(rb, db, hanning_window(h, w))


def _cv_round(x: np.ndarray) -> np.ndarray:
    return np.rint(x).astype(np.int64)          # round half to even, as cvRound / saturate_cast<int>(double)


def _reflect(p: np.ndarray, n: int) -> np.ndarray:
    """cv::borderInterpolate BORDER_REFLECT (fedcba|abcdefgh|hgfedcb)."""
    if n == 1:
        return np.zeros_like(p)
    p = p.copy()
    for _ in range(8):
        lo = p < 0
        hi = p >= n
        if not (lo.any() or hi.any()):
            break
        p[lo] = -p[lo] - 1
        p[hi] = 2 * n - 1 - p[hi]
    return p


def _affine_source_coords(Minv: np.ndarray, h: int, w: int):
    """Fixed-point source coordinates of cv::warpAffine (WarpAffineInvoker): integer part and 1/32 fractions."""
    m = Minv.astype(np.float64)
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = _cv_round(m[0, 0] * xs * AB_SCALE)
    bdelta = _cv_round(m[1, 0] * xs * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = _cv_round((m[0, 1] * ys + m[0, 2]) * AB_SCALE) + round_delta
    Y0 = _cv_round((m[1, 1] * ys + m[1, 2]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    return X >> INTER_BITS, Y >> INTER_BITS, X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def invert_affine(M: np.ndarray) -> np.ndarray:
    """cv::invertAffineTransform in double."""
    m = M.astype(np.float64)
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = m[1, 1] * D, m[0, 0] * D, -m[0, 1] * D, -m[1, 0] * D
    b1 = -A11 * m[0, 2] - A12 * m[1, 2]
    b2 = -A21 * m[0, 2] - A22 * m[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]], np.float64)


def warp_affine(src: np.ndarray, M: np.ndarray, inverse_map: bool, border: str = "reflect", nearest: bool = False) -> np.ndarray:
    """cv2.warpAffine(src, M, (w, h), INTER_LINEAR [| WARP_INVERSE_MAP], BORDER_REFLECT or BORDER_CONSTANT(0)).
    uint8 (any channel count) or float32 single channel; `nearest`: INTER_NEAREST (uint8 masks)."""
    h, w = src.shape[:2]
    Minv = M.astype(np.float64) if inverse_map else invert_affine(M)
    if nearest:
        m = Minv
        xs = np.arange(w, dtype=np.float64)
        ys = np.arange(h, dtype=np.float64)
        adelta = _cv_round(m[0, 0] * xs * AB_SCALE)
        bdelta = _cv_round(m[1, 0] * xs * AB_SCALE)
        X0 = _cv_round((m[0, 1] * ys + m[0, 2]) * AB_SCALE) + AB_SCALE // 2
        Y0 = _cv_round((m[1, 1] * ys + m[1, 2]) * AB_SCALE) + AB_SCALE // 2
        sx = (X0[:, None] + adelta[None, :]) >> AB_BITS
        sy = (Y0[:, None] + bdelta[None, :]) >> AB_BITS
        inside = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        out = np.zeros_like(src)
        out[inside] = src[sy[inside], sx[inside]]
        return out
    sx, sy, ax, ay = _affine_source_coords(Minv, h, w)
    x0, x1, y0, y1 = sx, sx + 1, sy, sy + 1
    if border == "reflect":
        valid = None
        x0, x1, y0, y1 = _reflect(x0, w), _reflect(x1, w), _reflect(y0, h), _reflect(y1, h)
    else:
        valid = [(x0 >= 0) & (x0 < w), (x1 >= 0) & (x1 < w), (y0 >= 0) & (y0 < h), (y1 >= 0) & (y1 < h)]
        x0, x1, y0, y1 = np.clip(x0, 0, w - 1), np.clip(x1, 0, w - 1), np.clip(y0, 0, h - 1), np.clip(y1, 0, h - 1)

    def tap(yy, xx, vy, vx):
        v = src[yy, xx]
        if valid is not None:
            ok = valid[vy] & valid[vx]
            v = np.where(ok[..., None] if v.ndim == 3 else ok, v, 0)
        return v

    p00, p01, p10, p11 = tap(y0, x0, 2, 0), tap(y0, x1, 2, 1), tap(y1, x0, 3, 0), tap(y1, x1, 3, 1)
    if src.dtype == np.uint8:
        # BilinearTab_i: weights are multiples of 1/1024, exact in 15-bit fixed point
        w00 = ((INTER_TAB_SIZE - ax) * (INTER_TAB_SIZE - ay) * 32).astype(np.int64)
        w01 = (ax * (INTER_TAB_SIZE - ay) * 32).astype(np.int64)
        w10 = ((INTER_TAB_SIZE - ax) * ay * 32).astype(np.int64)
        w11 = (ax * ay * 32).astype(np.int64)
        if src.ndim == 3:
            w00, w01, w10, w11 = w00[..., None], w01[..., None], w10[..., None], w11[..., None]
        acc = p00.astype(np.int64) * w00 + p01.astype(np.int64) * w01 + p10.astype(np.int64) * w10 + p11.astype(np.int64) * w11
        return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    fx = (ax.astype(np.float32) / np.float32(INTER_TAB_SIZE))
    fy = (ay.astype(np.float32) / np.float32(INTER_TAB_SIZE))
    one = np.float32(1.0)
    w00, w01, w10, w11 = (one - fx) * (one - fy), fx * (one - fy), (one - fx) * fy, fx * fy
    return (p00.astype(np.float32) * w00 + p01.astype(np.float32) * w01 + p10.astype(np.float32) * w10 + p11.astype(np.float32) * w11).astype(np.float32)


def _filter_dx(img: np.ndarray, axis: int) -> np.ndarray:
    """cv2.filter2D with Matx13f(-0.5, 0, 0.5) (or its transpose), BORDER_REFLECT_101."""
    p = np.pad(img, 1, mode="reflect")
    if axis == 1:
        return (np.float32(0.5) * p[1:-1, 2:] - np.float32(0.5) * p[1:-1, :-2]).astype(np.float32)
    return (np.float32(0.5) * p[2:, 1:-1] - np.float32(0.5) * p[:-2, 1:-1]).astype(np.float32)


def find_transform_ecc_euclidean(template: np.ndarray, image: np.ndarray, mask_u8: Optional[np.ndarray], iters: int, eps: float):
    """cv2.findTransformECC(template, image, eye(2,3), MOTION_EUCLIDEAN, (EPS|COUNT, iters, eps), inputMask, gaussFiltSize=1).
    Returns (rho, warp float32 2x3)."""
    hs, ws = template.shape
    tpl = template.astype(np.float32)
    img = image.astype(np.float32)
    pre = np.ones((hs, ws), np.uint8) if mask_u8 is None else (mask_u8 > 0).astype(np.uint8)
    pre_f = pre.astype(np.float32) * np.float32(0.5 / 0.95)
    pre = np.rint(pre_f).astype(np.uint8)        # convertTo rounds: 0.526 -> 1
    pre_f = pre.astype(np.float32)
    gx = _filter_dx(img, 1) * pre_f
    gy = _filter_dx(img, 0) * pre_f
    xg = np.tile(np.arange(ws, dtype=np.float32)[None, :], (hs, 1))
    yg = np.tile(np.arange(hs, dtype=np.float32)[:, None], (1, ws))
    warp = np.eye(2, 3, dtype=np.float32)
    rho, last_rho = -1.0, -float(eps)
    it = 0
    while it < iters and abs(rho - last_rho) >= eps:
        it += 1
        iw = warp_affine(img, warp, True, border="constant")
        gxw = warp_affine(gx, warp, True, border="constant")
        gyw = warp_affine(gy, warp, True, border="constant")
        im_mask = warp_affine(pre, warp, True, nearest=True) > 0
        cnt = int(im_mask.sum())
        iv = iw[im_mask].astype(np.float64)
        tv = tpl[im_mask].astype(np.float64)
        img_mean, img_std = iv.mean(), iv.std()
        tmp_mean, tmp_std = tv.mean(), tv.std()
        iwz = np.where(im_mask, (iw - np.float32(img_mean)).astype(np.float32), iw)      # subtract(..., mask): outside unchanged
        tz = np.where(im_mask, (tpl - np.float32(tmp_mean)).astype(np.float32), np.float32(0))
        tmp_norm = math.sqrt(cnt * tmp_std * tmp_std)
        img_norm = math.sqrt(cnt * img_std * img_std)
        h0, h1 = warp[0, 0], warp[1, 0]
        hat_x = -(xg * h1) - (yg * h0)
        hat_y = (xg * h0) - (yg * h1)
        j0 = gxw * hat_x + gyw * hat_y
        J = [j0, gxw, gyw]
        hess = np.array([[float(np.dot(J[a].ravel().astype(np.float64), J[b].ravel().astype(np.float64))) for b in range(3)] for a in range(3)],
                        np.float32)
        hinv = np.linalg.inv(hess.astype(np.float64)).astype(np.float32)
        corr = float(np.dot(tz.ravel().astype(np.float64), iwz.ravel().astype(np.float64)))
        last_rho = rho
        rho = corr / (img_norm * tmp_norm)
        if not np.isfinite(rho):
            raise RuntimeError("ECC: NaN correlation")
        proj = lambda v: np.array([float(np.dot(J[a].ravel().astype(np.float64), v.ravel().astype(np.float64))) for a in range(3)], np.float32)
        ip, tp = proj(iwz), proj(tz)
        iph = (hinv @ ip).astype(np.float32)
        lam_n = img_norm * img_norm - float(np.dot(ip.astype(np.float64), iph.astype(np.float64)))
        lam_d = corr - float(np.dot(tp.astype(np.float64), iph.astype(np.float64)))
        if lam_d <= 0.0:
            raise RuntimeError("ECC stopped before convergence (lambda_d <= 0)")
        lam = lam_n / lam_d
        err = (np.float32(lam) * tz - iwz).astype(np.float32)
        dp = (hinv @ proj(err)).astype(np.float32)
        theta = np.float32(math.asin(float(warp[1, 0]))) + dp[0]
        warp[0, 2] += dp[1]
        warp[1, 2] += dp[2]
        warp[0, 0] = warp[1, 1] = np.float32(math.cos(float(theta)))
        warp[1, 0] = np.float32(math.sin(float(theta)))
        warp[0, 1] = -warp[1, 0]
    return rho, warp, it


def align_crop_ecc(ref_u8: np.ndarray, mov_u8: np.ndarray, mask_bool: Optional[np.ndarray], iters=300, eps=1e-7, gauss_filt=5):
    """shape_ftp.py:549-578 (mode "euclidean")."""
    ref = ref_u8.astype(np.float32) / np.float32(255.0)
    mov = mov_u8.astype(np.float32) / np.float32(255.0)
    if gauss_filt and gauss_filt > 0:
        ref = cvlite.gaussian_blur(ref, float(gauss_filt))
        mov = cvlite.gaussian_blur(mov, float(gauss_filt))
    m = None if mask_bool is None else (mask_bool.astype(np.uint8) * 255)
    try:
        rho, warp, n_it = find_transform_ecc_euclidean(ref, mov, m, int(iters), float(eps))
    except RuntimeError:
        # cv2.error in the reference: the unaligned crop and the identity warp are used (shape_ftp.py:576-578)
        return mov_u8, np.eye(2, 3, dtype=np.float32), float("nan"), 0
    aligned = warp_affine(mov_u8, warp, True, border="reflect")
    return aligned, warp, float(rho), n_it


def aligned_crops(reference_path: str, deformed_path: str, circle_pts, apply_global_shift=True, use_ecc=True,
                  ecc_iters=300, ecc_eps=1e-7, ecc_gauss=5, gray_generation: int = 4):
    """shape_ftp.main :1471-1537: returns (ref_gray crop u8, aligned deformed gray crop u8, (cx, cy, r) local, info)."""
    from . import ftp_oracle as O
    ref_bgr = imread_bgr(reference_path)
    def_bgr = imread_bgr(deformed_path)
    H, W = ref_bgr.shape[:2]
    shift, response = estimate_global_shift(bgr2gray_u8(ref_bgr, gray_generation).astype(np.float32), bgr2gray_u8(def_bgr, gray_generation).astype(np.float32))
    if apply_global_shift:
        M = np.array([[1, 0, shift[0]], [0, 1, shift[1]]], np.float32)
        def_bgr = warp_affine(def_bgr, M, False, border="reflect")
    cx, cy, r = O.circle_from_3_points(*circle_pts)
    x1, x2, y1, y2 = max(0, cx - r), min(W, cx + r), max(0, cy - r), min(H, cy + r)
    ref_gray = bgr2gray_u8(ref_bgr[y1:y2, x1:x2], gray_generation)
    def_gray = bgr2gray_u8(def_bgr[y1:y2, x1:x2], gray_generation)
    h, w = ref_gray.shape
    cxl, cyl = cx - x1, cy - y1
    rl = int(min(r, cxl, cyl, w - 1 - cxl, h - 1 - cyl))
    info = {"shift": shift, "response": response, "crop": (x1, x2, y1, y2)}
    if use_ecc:
        circ = O.circular_mask(h, w, cxl, cyl, rl)
        def_gray, warp, rho, n_it = align_crop_ecc(ref_gray, def_gray, circ, ecc_iters, ecc_eps, ecc_gauss)
        info.update(warp=warp, rho=rho, ecc_iters=n_it)
    return ref_gray, def_gray, (cxl, cyl, rl), info
