"""Pre-path alignment on the GPU: host mirror of `Code/shape_ftp.py:1471-1537`.

`FtpAligner(reference_bgr, circle_points=...)` does once what `shape_ftp.main` does with the reference photograph
(BGR2GRAY, blurred + windowed spectrum for `estimate_global_shift`, ROI crop, ECC template);
`align(deformed_bgr)` returns the aligned grey ROI crops the FTP path consumes, plus, per frame, the phase-correlation
shift, the ECC warp, rho and the iteration count.  Image decoding stays on the host (the reference uses cv2.imread).
All arithmetic runs in libvistaf_ftp.so (include/vistaf_align.h); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

NINFO = 12


class AlignConfig(ctypes.Structure):
    _fields_ = [("apply_global_shift", ctypes.c_int32), ("use_ecc", ctypes.c_int32), ("ecc_iters", ctypes.c_int32), ("gray_coeffs", ctypes.c_int32),
                ("ecc_eps", ctypes.c_double), ("ecc_gauss_sigma", ctypes.c_double), ("shift_blur_sigma", ctypes.c_double)]


def circle_from_3_points(p1, p2, p3) -> Tuple[int, int, int]:
    """shape_ftp.circle_from_3_points (:388-414): circumcircle, centre and radius rounded to int."""
    (x1, y1), (x2, y2), (x3, y3) = [(float(a), float(b)) for a, b in (p1, p2, p3)]
    d = 2.0 * (x1 * (y2 - y3) + x2 * (y3 - y1) + x3 * (y1 - y2))
    if abs(d) < 1e-12:
        raise ValueError("The 3 points are collinear; cannot define a circle.")
    ux = ((x1 ** 2 + y1 ** 2) * (y2 - y3) + (x2 ** 2 + y2 ** 2) * (y3 - y1) + (x3 ** 2 + y3 ** 2) * (y1 - y2)) / d
    uy = ((x1 ** 2 + y1 ** 2) * (x3 - x2) + (x2 ** 2 + y2 ** 2) * (x1 - x3) + (x3 ** 2 + y3 ** 2) * (x2 - x1)) / d
    r = float(np.sqrt((x1 - ux) ** 2 + (y1 - uy) ** 2))
    return int(round(ux)), int(round(uy)), int(round(r))


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class FtpAligner:
    """Session for the alignment of deformed photographs to one reference photograph (full frames, uint8 BGR)."""

    def __init__(self, reference_bgr, circle: Optional[Tuple[int, int, int]] = None,
                 circle_points: Sequence[Tuple[int, int]] = ((1873, 1703), (1599, 707), (2575, 950)),      # shape_ftp.py:41-43
                 apply_global_shift: bool = True, use_ecc: bool = True, ecc_iters: int = 300, ecc_eps: float = 1e-7,
                 ecc_gauss_sigma: float = 5.0, max_batch: int = 1, device=None, gray_coeffs: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("FtpAligner needs a HIP device (there is no CPU fallback)")
        self._lib = _lib.load()
        self._lib.vistaf_align_create.restype = ctypes.c_int
        self.device = torch.device(device if device is not None else "cuda:0")
        ref = self._as_bgr(reference_bgr)
        if ref.shape[0] != 1:
            raise ValueError("one reference frame expected")
        self.H, self.W = int(ref.shape[1]), int(ref.shape[2])
        self.circle_full = tuple(int(v) for v in (circle if circle is not None else circle_from_3_points(*circle_points)))
        cfg = AlignConfig()
        self._lib.vistaf_align_default_config(ctypes.byref(cfg))
        cfg.apply_global_shift = int(apply_global_shift); cfg.use_ecc = int(use_ecc); cfg.ecc_iters = int(ecc_iters)
        cfg.ecc_eps = float(ecc_eps); cfg.ecc_gauss_sigma = float(ecc_gauss_sigma)
        cfg.gray_coeffs = int(gray_coeffs)       # 0: OpenCV 4.x BGR2GRAY coefficients, 1: OpenCV 3.x (include/vistaf_align.h)
        self.max_batch = int(max_batch)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_align_create(ctypes.byref(cfg), self.H, self.W, *self.circle_full, self.max_batch, ctypes.byref(self._h)))
        g = [ctypes.c_int32() for _ in range(9)]
        _lib.check(self._lib.vistaf_align_geometry(self._h, *[ctypes.byref(v) for v in g]))
        self.crop_box = (g[0].value, g[1].value, g[2].value, g[3].value)              # x1, y1, x2, y2
        self.crop_shape = (g[4].value, g[5].value)
        self.circle_crop = (g[6].value, g[7].value, g[8].value)
        self.reference_gray_crop = torch.empty(self.crop_shape, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_align_set_reference(self._h, ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(self.reference_gray_crop.data_ptr()),
                                                             _stream_ptr(self.device)))

    def _as_bgr(self, frames) -> torch.Tensor:
        t = torch.as_tensor(frames)
        if t.dim() == 3:
            t = t[None]
        if t.dim() != 4 or t.shape[-1] != 3 or t.dtype != torch.uint8:
            raise ValueError("frames must be [B,H,W,3] or [H,W,3] uint8 BGR")
        return t.to(self.device).contiguous()

    def align(self, deformed_bgr) -> Dict[str, Any]:
        """-> {"aligned_gray": [B,h,w] uint8 (device), "info": [B,12] float64 (host), "shift", "warp", "rho", "ecc_iters", "ecc_failed"}"""
        t = self._as_bgr(deformed_bgr)
        b = int(t.shape[0])
        if b > self.max_batch:
            raise ValueError("batch exceeds max_batch")
        if int(t.shape[1]) != self.H or int(t.shape[2]) != self.W:
            raise RuntimeError("Reference and deformed images have different sizes.")        # shape_ftp.py:1479-1480
        out = torch.empty((b,) + self.crop_shape, dtype=torch.uint8, device=self.device)
        info = torch.empty((b, NINFO), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_align_batch(self._h, ctypes.c_void_p(t.data_ptr()), b, ctypes.c_void_p(out.data_ptr()),
                                                     ctypes.c_void_p(info.data_ptr()), _stream_ptr(self.device)))
        hi = info.cpu().numpy()
        return {"aligned_gray": out, "info": hi, "shift": hi[:, 0:2].copy(), "response": hi[:, 2].copy(), "warp": hi[:, 3:9].reshape(b, 2, 3).copy(),
                "rho": hi[:, 9].copy(), "ecc_iters": hi[:, 10].astype(int), "ecc_failed": hi[:, 11].astype(bool)}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.vistaf_align_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
