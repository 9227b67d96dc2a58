"""ctypes wrapper over oracle/cvlite.c -- TEST INFRASTRUCTURE ONLY (the parity oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "cvlite.c")
_LIB = os.path.join(_HERE, "_build", "libcvlite.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile cvlite.c with gcc into oracle/_build/libcvlite.so (idempotent)."""
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    if force or (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-o", _LIB, _SRC, "-lm"]
        )
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.cvl_gaussian_ksize_f32.restype = ctypes.c_int
        _lib.cvl_gaussian_ksize_f32.argtypes = [ctypes.c_double]
        _lib.cvl_cc8_label.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def gaussian_ksize(sigma: float) -> int:
    return int(lib().cvl_gaussian_ksize_f32(float(sigma)))


def gaussian_kernel(n: int, sigma: float) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().cvl_gaussian_kernel_f32(ctypes.c_int(n), ctypes.c_double(sigma), _p(out))
    return out


def gaussian_blur(src, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(src_f32, (0, 0), sigma) -- BORDER_REFLECT_101."""
    src = _f32(src)
    dst = np.empty_like(src)
    h, w = src.shape
    lib().cvl_gaussian_blur_f32(_p(src), _p(dst), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_double(sigma))
    return dst


def sobel3(src):
    """(cv2.Sobel(src, CV_32F, 1, 0, ksize=3), cv2.Sobel(src, CV_32F, 0, 1, ksize=3))."""
    src = _f32(src)
    gx = np.empty_like(src)
    gy = np.empty_like(src)
    h, w = src.shape
    lib().cvl_sobel3_f32(_p(src), _p(gx), _p(gy), ctypes.c_int(h), ctypes.c_int(w))
    return gx, gy


def ellipse_se(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k))."""
    se = np.empty((k, k), np.uint8)
    lib().cvl_ellipse_se(ctypes.c_int(k), _p(se))
    return se


def _morph(fn, src, se, iters):
    src = _u8(src)
    se = _u8(se)
    dst = np.empty_like(src)
    h, w = src.shape
    fn(_p(src), _p(dst), ctypes.c_int(h), ctypes.c_int(w), _p(se), ctypes.c_int(se.shape[0]), ctypes.c_int(iters))
    return dst


def dilate(src_u8, se, iters: int = 1) -> np.ndarray:
    return _morph(lib().cvl_dilate_u8, src_u8, se, iters)


def erode(src_u8, se, iters: int = 1) -> np.ndarray:
    return _morph(lib().cvl_erode_u8, src_u8, se, iters)


def morph_close(src_u8, se, iters: int = 1) -> np.ndarray:
    """cv2.morphologyEx(src, MORPH_CLOSE, se, iterations=iters) = erode(dilate(src))."""
    return erode(dilate(src_u8, se, iters), se, iters)


def cc8(mask):
    """cv2.connectedComponentsWithStats(mask, connectivity=8) -> (num, labels, areas)."""
    src = _u8(mask)
    h, w = src.shape
    labels = np.empty((h, w), np.int32)
    cap = h * w // 1 + 2
    areas = np.zeros(cap, np.int32)
    num = lib().cvl_cc8_label(_p(src), _p(labels), _p(areas), ctypes.c_int(cap), ctypes.c_int(h), ctypes.c_int(w))
    return int(num), labels, areas[:num].copy()


def dist_l2_3x3(src_u8) -> np.ndarray:
    """cv2.distanceTransform(src, cv2.DIST_L2, 3)."""
    src = _u8(src_u8)
    h, w = src.shape
    dst = np.empty((h, w), np.float32)
    lib().cvl_dist_l2_3x3(_p(src), _p(dst), ctypes.c_int(h), ctypes.c_int(w))
    return dst


def box_sum(src, k: int) -> np.ndarray:
    """cv2.boxFilter(src, -1, (k, k), normalize=False)."""
    src = _f32(src)
    dst = np.empty_like(src)
    h, w = src.shape
    lib().cvl_box_sum_f32(_p(src), _p(dst), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(k))
    return dst


def pad_reflect(src, pad: int) -> np.ndarray:
    """cv2.copyMakeBorder(src, pad, pad, pad, pad, cv2.BORDER_REFLECT)."""
    src = _f32(src)
    h, w = src.shape
    dst = np.empty((h + 2 * pad, w + 2 * pad), np.float32)
    lib().cvl_pad_reflect_f32(_p(src), _p(dst), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(pad))
    return dst


def inpaint_telea(src, mask, radius: float) -> np.ndarray:
    """cv2.inpaint(src_f32, mask_u8, radius, cv2.INPAINT_TELEA)."""
    src = _f32(src)
    mask = _u8(mask)
    h, w = src.shape
    dst = np.empty_like(src)
    lib().cvl_inpaint_telea_f32(_p(src), _p(mask), _p(dst), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_double(radius))
    return dst


def inpaint_telea_u8(src_u8, mask, radius: float) -> np.ndarray:
    """cv2.inpaint(src_u8, mask_u8, radius, cv2.INPAINT_TELEA) on an 8-bit single-channel image (parity unpinned: no file of the reference holds
    an output of this call)."""
    src = _f32(np.asarray(src_u8, np.uint8))
    m = _u8(mask)
    dst = np.empty_like(src)
    h, w = src.shape
    lib().cvl_inpaint_telea_u8_f32(_p(src), _p(m), _p(dst), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_double(radius))
    return dst.astype(np.uint8)


def gaussian_blur_xy(src, sigma_x: float, sigma_y: float) -> np.ndarray:
    """cv2.GaussianBlur(src_f32, (0, 0), sigmaX=sigma_x, sigmaY=sigma_y) -- BORDER_REFLECT_101."""
    src = _f32(src)
    dst = np.empty_like(src)
    h, w = src.shape
    lib().cvl_gaussian_blur_xy_f32(_p(src), _p(dst), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_double(sigma_x), ctypes.c_double(sigma_y))
    return dst


def inpaint_telea_order(src, mask, radius: float):
    """inpaint_telea plus the march's fill sequence: (dst, fill_index int32 [h, w], -1 where nothing was filled)."""
    src = _f32(src)
    m = _u8(mask)
    dst = np.empty_like(src)
    order = np.empty(src.shape, np.int32)
    h, w = src.shape
    f = lib().cvl_inpaint_telea_f32_order
    f.restype = ctypes.c_int
    f(_p(src), _p(m), _p(dst), _p(order), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_double(radius))
    return dst, order


def inpaint_telea_pops(src, mask, radius: float, cap: int = 1 << 20):
    """inpaint_telea plus the pop log of its two FMM passes: (dst, rows int32 [n, 3] = (pass, i, j) in padded coordinates, T float32 [n])."""
    src = _f32(src)
    m = _u8(mask)
    dst = np.empty_like(src)
    log = np.empty((cap, 3), np.int32)
    tt = np.empty(cap, np.float32)
    h, w = src.shape
    f = lib().cvl_inpaint_telea_f32_pops
    f.restype = ctypes.c_int
    n = f(_p(src), _p(m), _p(dst), _p(log), _p(tt), ctypes.c_int(cap), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_double(radius))
    n = min(n, cap)
    return dst, log[:n].copy(), tt[:n].copy()


def unwrap_quality_guided(wrapped, mask, quality, want_tree: bool = False):
    """shape_ftp.unwrap_quality_guided (shape_ftp.py:1043-1080), exact heap order."""
    wrapped = _f32(wrapped)
    quality = _f32(quality)
    mask = _u8(mask)
    h, w = wrapped.shape
    out = np.empty((h, w), np.float32)
    parent = np.empty((h, w), np.int32) if want_tree else None
    order = np.empty((h, w), np.int32) if want_tree else None
    lib().cvl_unwrap_quality_guided(
        _p(wrapped), _p(mask), _p(quality), _p(out),
        _p(parent) if want_tree else None, _p(order) if want_tree else None,
        ctypes.c_int(h), ctypes.c_int(w),
    )
    if want_tree:
        return out, parent, order
    return out
