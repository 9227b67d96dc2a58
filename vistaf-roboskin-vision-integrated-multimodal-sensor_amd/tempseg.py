"""Host-side mirror of the reference's periodic-stripe segmentation (first slice of the temperature modality).

Reference interface mirrored: Code/temperature_sensor.py
  * `segment_dark_light_gratings_periodic_fft(image_bgr, roi_full) -> (dark_final, light_final, pack)`  (:437-540)
  * `circle_from_three_points`, `roi_mask_from_circle`, `bbox_from_mask`, `crop2d`                        (:156-216)
  * `compute_feature_planes(image_bgr, blur_ksize) -> {"L", "a", "b", "gray"}`                            (:278-293)
  * the chroma / colour-support test of `main()` (with `dilate_bool_mask`, :583-590)                       (:793-799)
backed by `vistaf_tempseg_*` of libvistaf_ftp.so (include/vistaf_temp.h).  PyTorch only holds the device buffers.  The temperature
regressors (`TempModel.predict`, :236) are not part of this slice: their parameters only exist as pickled scikit-learn pipelines.
"""
from __future__ import annotations

import ctypes
import dataclasses
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib


@dataclasses.dataclass
class TempSegConfig:
    """Code/temperature_sensor.py:67-82, as shipped"""
    seg_band_radius: int = 22
    seg_dc_exclusion: int = 28
    seg_illum_sigma: int = 20
    sat_thresh_gray: int = 245
    sat_dilate_ksize: int = 13
    post_close_kx: int = 3
    post_close_ky: int = 31
    post_open_kx: int = 3
    post_open_ky: int = 7
    n_peaks: int = 16
    seg_peak_max_dy_from_center: float = 0.14

    def to_c(self) -> "_lib.CTempSegConfig":
        cc = _lib.CTempSegConfig()
        for f in dataclasses.fields(self):
            setattr(cc, f.name, getattr(self, f.name))
        return cc


OUTER_CIRCLE = ((1845, 1818), (1517, 623), (2687, 914))       # temperature_sensor.py:37-39
CROP_PAD_PX = 10                                              # :49
BLUR_KSIZE = 5                                                # :52
COLOR_CHROMA_MIN = 10.0                                       # :86
COLOR_SUPPORT_DILATE = 3                                      # :87
COLOR_T_MIN, COLOR_T_MAX = 20.0, 33.0                         # :55-56
COLOR_GUARD_BAND, SWITCH_MARGIN_C = 0.5, 1.0                  # :59-60
FINAL_T_MIN, FINAL_T_MAX = 20.0, 75.0                         # :63-64
FINAL_SMOOTH_SIGMA_ACROSS, FINAL_SMOOTH_SIGMA_ALONG = 6.0, 1.0   # :95-96


def circle_from_three_points(p1, p2, p3, eps: float = 1e-12) -> Tuple[float, float, float]:
    """temperature_sensor.circle_from_three_points (:156-177)"""
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
    """temperature_sensor.roi_mask_from_circle (:180-184)"""
    cx, cy, r = circle_from_three_points(p1, p2, p3)
    yy, xx = np.ogrid[:h, :w]
    return (xx - cx) ** 2 + (yy - cy) ** 2 <= r ** 2


def bbox_from_mask(mask: np.ndarray, pad: int = 0) -> Tuple[int, int, int, int]:
    """temperature_sensor.bbox_from_mask (:194-208): (y0, y1, x0, x1), ends exclusive; the full frame for an empty mask"""
    h, w = mask.shape[:2]
    ys, xs = np.where(mask)
    if ys.size == 0:
        return 0, h, 0, w
    return (int(max(0, ys.min() - int(pad))), int(min(h, ys.max() + int(pad) + 1)), int(max(0, xs.min() - int(pad))),
            int(min(w, xs.max() + int(pad) + 1)))


def crop2d(arr: np.ndarray, bbox: Optional[Tuple[int, int, int, int]]) -> np.ndarray:
    """temperature_sensor.crop2d (:211-216)"""
    if bbox is None:
        return arr
    y0, y1, x0, x1 = bbox
    return arr[y0:y1, x0:x1]


class TempSegmenter:
    """One segmentation session for H x W photographs (workspace, hipFFT plan and tables allocated once)."""

    def __init__(self, H: int, W: int, config: Optional[TempSegConfig] = None, device="cuda:0"):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        if not torch.cuda.is_available():
            raise RuntimeError("TempSegmenter needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device)
        self.config = config or TempSegConfig()
        self.H, self.W = int(H), int(W)
        cc = self.config.to_c()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_tempseg_create(ctypes.byref(cc), self.H, self.W, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.vistaf_tempseg_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def segment(self, image_bgr, roi_full) -> Tuple[np.ndarray, np.ndarray, Dict[str, Any]]:
        """segment_dark_light_gratings_periodic_fft(image_bgr, roi_full) -> (dark_final, light_final, pack); pack holds the reference's
        "dbg", "roi_eff", "sat", "peak", "angle_rad" and "period_px" entries (not the float planes "fft_mag" / "signal", which only feed
        debug figures upstream)."""
        img = image_bgr if torch.is_tensor(image_bgr) else torch.from_numpy(np.ascontiguousarray(image_bgr))
        roi = roi_full if torch.is_tensor(roi_full) else torch.from_numpy(np.ascontiguousarray(np.asarray(roi_full)).astype(np.uint8))
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or tuple(img.shape[:2]) != (self.H, self.W):
            raise ValueError(f"image must be [{self.H},{self.W},3] uint8 (BGR)")
        if tuple(roi.shape) != (self.H, self.W):
            raise ValueError("roi mask shape does not match the image")
        img = img.to(self.device).contiguous()
        roi = (roi.to(self.device) != 0).to(torch.uint8).contiguous()
        outs = [torch.empty((self.H, self.W), dtype=torch.uint8, device=self.device) for _ in range(4)]
        info = (ctypes.c_double * _lib.TEMPSEG_NINFO)()
        stream = int(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_tempseg_segment(self._h, img.data_ptr(), roi.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                                        outs[2].data_ptr(), outs[3].data_ptr(), info, stream))
        dark, light, roi_eff, sat = (o.cpu().numpy().astype(bool) for o in outs)
        dbg = {
            "peak_x": int(info[0]), "peak_y": int(info[1]), "phi0_rad": float(info[2]), "mean_gray_A": float(info[3]), "mean_gray_B": float(info[4]),
            "chosen": "A_is_dark" if info[5] else "B_is_dark", "roi_pixels": int(info[6]), "roi_eff_pixels": int(info[7]), "sat_pixels": int(info[8]),
            "dark_pixels": int(info[9]), "light_pixels": int(info[10]), "carrier_angle_rad": float(info[11]), "carrier_period_px": float(info[12]),
        }
        pack = {"dbg": dbg, "roi_eff": roi_eff, "sat": sat, "peak": (dbg["peak_x"], dbg["peak_y"]), "angle_rad": dbg["carrier_angle_rad"],
                "period_px": dbg["carrier_period_px"]}
        return dark, light, pack


    # ---- second slice: feature planes and colour support -----------------------------------------------------------------------------
    def _dev_u8(self, m, what):
        t = m if torch.is_tensor(m) else torch.from_numpy(np.ascontiguousarray(np.asarray(m)).astype(np.uint8))
        if tuple(t.shape) != (self.H, self.W):
            raise ValueError(f"{what} mask shape does not match the frame")
        return (t.to(self.device) != 0).to(torch.uint8).contiguous()

    def feature_planes_device(self, image_bgr, blur_ksize: int = BLUR_KSIZE) -> Dict[str, torch.Tensor]:
        """compute_feature_planes (:278-293) with the planes left on the device (float32 [H, W] tensors "L", "a", "b", "gray")"""
        img = image_bgr if torch.is_tensor(image_bgr) else torch.from_numpy(np.ascontiguousarray(image_bgr))
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or tuple(img.shape[:2]) != (self.H, self.W):
            raise ValueError(f"image must be [{self.H},{self.W},3] uint8 (BGR)")
        img = img.to(self.device).contiguous()
        planes = {k: torch.empty((self.H, self.W), dtype=torch.float32, device=self.device) for k in ("L", "a", "b", "gray")}
        stream = int(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_temp_feature_planes(self._h, img.data_ptr(), int(blur_ksize), planes["L"].data_ptr(), planes["a"].data_ptr(),
                                                            planes["b"].data_ptr(), planes["gray"].data_ptr(), stream))
        return planes

    def feature_planes(self, image_bgr, blur_ksize: int = BLUR_KSIZE) -> Dict[str, np.ndarray]:
        """Drop-in for temperature_sensor.compute_feature_planes(image_bgr, blur_ksize) (:278-293): float32 planes "L", "a", "b", "gray"."""
        return {k: v.cpu().numpy() for k, v in self.feature_planes_device(image_bgr, blur_ksize).items()}

    def color_support(self, planes, light_mask, roi_eff, sat, chroma_min: float = COLOR_CHROMA_MIN, dilate_ksize: int = COLOR_SUPPORT_DILATE):
        """main() :793-799 -> (color_support bool [H, W], chroma float32 [H, W]); `planes` as returned by feature_planes[_device]"""
        a, b = (p if torch.is_tensor(p) else torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)) for p in (planes["a"], planes["b"]))
        a, b = a.to(self.device).contiguous(), b.to(self.device).contiguous()
        if a.dtype != torch.float32 or b.dtype != torch.float32 or tuple(a.shape) != (self.H, self.W) or tuple(b.shape) != (self.H, self.W):
            raise ValueError("planes must be float32 [H, W]")
        light, roi_e, sat_m = self._dev_u8(light_mask, "light"), self._dev_u8(roi_eff, "roi_eff"), self._dev_u8(sat, "sat")
        chroma = torch.empty((self.H, self.W), dtype=torch.float32, device=self.device)
        support = torch.empty((self.H, self.W), dtype=torch.uint8, device=self.device)
        stream = int(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_temp_color_support(self._h, a.data_ptr(), b.data_ptr(), light.data_ptr(), roi_e.data_ptr(), sat_m.data_ptr(),
                                                           float(chroma_min), int(dilate_ksize), chroma.data_ptr(), support.data_ptr(), stream))
        return support.cpu().numpy().astype(bool), chroma.cpu().numpy()

    # ---- third slice: the map-domain stages behind the regressors (parity unpinned, see include/vistaf_temp.h) ------------------------
    def _dev_f32(self, m, what):
        t = m if torch.is_tensor(m) else torch.from_numpy(np.ascontiguousarray(np.asarray(m), dtype=np.float32))
        if tuple(t.shape) != (self.H, self.W) or t.dtype != torch.float32:
            raise ValueError(f"{what} must be float32 [H, W]")
        return t.to(self.device).contiguous()

    def _stream(self):
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    def clamp_map(self, m, roi, lo: float, hi: float) -> np.ndarray:
        """clamp_map (:538-543)"""
        mp, r = self._dev_f32(m, "map"), self._dev_u8(roi, "roi")
        out = torch.empty_like(mp)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_temp_clamp_map(self._h, mp.data_ptr(), r.data_ptr(), float(lo), float(hi), out.data_ptr(), self._stream()))
        return out.cpu().numpy()

    def inpaint_temperature_map(self, temp_map, roi_mask, radius: int = 7) -> np.ndarray:
        """inpaint_temperature_map (:546-580)"""
        mp, r = self._dev_f32(temp_map, "map"), self._dev_u8(roi_mask, "roi")
        out = torch.empty_like(mp)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_temp_inpaint_map(self._h, mp.data_ptr(), r.data_ptr(), int(radius), out.data_ptr(), self._stream()))
        return out.cpu().numpy()

    def fuse_maps_per_pixel(self, roi, wide_map, color_map, fuse_config: Optional["_lib.CTempFuseConfig"] = None):
        """fuse_maps_per_pixel (:594-636) -> (final float32, source uint8, dbg)"""
        r, wm, cm = self._dev_u8(roi, "roi"), self._dev_f32(wide_map, "wide map"), self._dev_f32(color_map, "colour map")
        fin = torch.empty_like(wm)
        src = torch.empty((self.H, self.W), dtype=torch.uint8, device=self.device)
        cfg = fuse_config or _lib.CTempFuseConfig(COLOR_T_MIN, COLOR_T_MAX, COLOR_GUARD_BAND, SWITCH_MARGIN_C, FINAL_T_MIN, FINAL_T_MAX)
        counts = (ctypes.c_int64 * 4)()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_temp_fuse_maps(self._h, r.data_ptr(), wm.data_ptr(), cm.data_ptr(), ctypes.byref(cfg), fin.data_ptr(), src.data_ptr(),
                                                       counts, self._stream()))
        dbg = {"roi_pixels": int(counts[0]), "wide_ok_pixels": int(counts[1]), "color_ok_pixels": int(counts[2]), "blend_pixels": int(counts[3])}
        return fin.cpu().numpy(), src.cpu().numpy(), dbg

    def oriented_gaussian_blur_float(self, map_f, roi, angle_rad: float, sigma_across: float, sigma_along: float) -> np.ndarray:
        """oriented_gaussian_blur_float (:705-747)"""
        mp, r = self._dev_f32(map_f, "map"), self._dev_u8(roi, "roi")
        out = torch.empty_like(mp)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_temp_oriented_blur(self._h, mp.data_ptr(), r.data_ptr(), float(angle_rad), float(sigma_across), float(sigma_along),
                                                           out.data_ptr(), self._stream()))
        return out.cpu().numpy()


_default: Optional[TempSegmenter] = None


def _session(h: int, w: int, cfg: Optional[TempSegConfig] = None) -> TempSegmenter:
    global _default
    cfg = cfg or (_default.config if _default is not None else TempSegConfig())
    if _default is None or (_default.H, _default.W) != (h, w) or _default.config != cfg:
        if _default is not None:
            _default.close()
        _default = TempSegmenter(h, w, cfg)
    return _default


def compute_feature_planes(image_bgr, blur_ksize: int = 5) -> Dict[str, np.ndarray]:
    """Drop-in for temperature_sensor.compute_feature_planes (:278)"""
    return _session(int(image_bgr.shape[0]), int(image_bgr.shape[1])).feature_planes(image_bgr, blur_ksize)


def color_support_mask(planes, light_mask, roi_eff, sat, chroma_min: float = 10.0, dilate_ksize: int = 3):
    """The colour-support test of temperature_sensor.main() (:793-799) -> (color_support, chroma)"""
    h, w = (int(v) for v in planes["a"].shape)
    return _session(h, w).color_support(planes, light_mask, roi_eff, sat, chroma_min, dilate_ksize)


def clamp_map(m, roi, lo: float, hi: float) -> np.ndarray:
    """Drop-in for temperature_sensor.clamp_map (:538)"""
    return _session(*(int(v) for v in np.shape(m))).clamp_map(m, roi, lo, hi)


def inpaint_temperature_map(temp_map, roi_mask, radius: int = 7) -> np.ndarray:
    """Drop-in for temperature_sensor.inpaint_temperature_map (:546)"""
    return _session(*(int(v) for v in np.shape(temp_map))).inpaint_temperature_map(temp_map, roi_mask, radius)


def fuse_maps_per_pixel(roi, wide_map, color_map):
    """Drop-in for temperature_sensor.fuse_maps_per_pixel (:594) with the constants as shipped (:55-64)"""
    return _session(*(int(v) for v in np.shape(wide_map))).fuse_maps_per_pixel(roi, wide_map, color_map)


def oriented_gaussian_blur_float(map_f, roi, angle_rad: float, sigma_across: float, sigma_along: float) -> np.ndarray:
    """Drop-in for temperature_sensor.oriented_gaussian_blur_float (:705)"""
    return _session(*(int(v) for v in np.shape(map_f))).oriented_gaussian_blur_float(map_f, roi, angle_rad, sigma_across, sigma_along)


def segment_dark_light_gratings_periodic_fft(image_bgr, roi_full, config: Optional[TempSegConfig] = None):
    """Drop-in for temperature_sensor.segment_dark_light_gratings_periodic_fft (:437): a session per frame size is built on first use."""
    return _session(int(image_bgr.shape[0]), int(image_bgr.shape[1]), config or TempSegConfig()).segment(image_bgr, roi_full)
