"""Host-side mirror of the reference's image -> height-map -> force interface, backed by the HIP library.

Reference interface mirrored (paths relative to the reference tree):
  * Code/shape_ftp.py:1428-2037  `main(reference_path, deformed_path, ..., return_results=True)` ->
    {"height_map_mm_crop", "roi_eroded_crop", "output_reliable_crop", "estimated_grating_period_px"}
  * Code/force_sensor.py:93-123  `depth_map_to_volume_cm3(height_map_mm, roi_mask, mm_per_px, depth_eps_mm)`
  * Code/force_sensor.py:149-167 `predict_force_from_volume(best_model, volume_cm3)`
  * Code/force_sensor.py:173-187 `estimate_mm_per_px(period)`
  * Code/shape_ftp.py:672-680 / force_sensor.py:142-147 calibration JSON loaders
`predict(image[, reference]) -> force_map` is the name BASELINE.json asks for; it does not exist
upstream and is defined here as shape_ftp.main's result dict plus the force tail of
Code/multimodal_sensor.py:388-419.

PyTorch is used only for device memory and streams; all image arithmetic runs in libvistaf_ftp.so.
"""
from __future__ import annotations

import ctypes
import json
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import FtpConfig

SCALAR_NAMES = [
    "volume_cm3", "contact_area_mm2", "max_depth_mm", "force_N", "argmax_depth_index", "estimated_grating_period_px",
    "mm_per_px", "min_height_unitless", "argmin_unitless_index", "reliable_count", "sign_flipped", "amp_threshold",
    "contact_threshold", "bg_median", "bad_pixels", "reserved",
]


def load_calibration(json_path: str) -> Tuple[Dict[str, Any], bool]:
    """shape_ftp.load_calibration (Code/shape_ftp.py:672-680): (best_model, use_negated_height)."""
    with open(json_path, "r", encoding="utf-8") as f:
        cal = json.load(f)
    return cal["best_model"], bool(cal.get("use_negated_height_for_fit", True))


def load_force_calibration(path: str) -> Dict[str, Any]:
    """force_sensor.load_force_calibration (Code/force_sensor.py:142-147)."""
    with open(path, "r", encoding="utf-8") as f:
        data = json.load(f)
    if "best_model" not in data:
        raise ValueError("Invalid force calibration JSON: missing 'best_model'")
    return data


def _curve(model: Dict[str, Any], what: str) -> _lib.Curve:
    t = model["type"]
    p = model["params"]
    if t not in _lib.CURVE_TYPES:
        raise ValueError(f"Unknown model type in {what}: {t}")
    c = _lib.Curve()
    c.type = _lib.CURVE_TYPES[t]
    if t == "poly2":
        c.a, c.b, c.c = float(p["c2"]), float(p["c1"]), float(p["c0"])
    else:
        c.a = float(p["a"])
        c.b = float(p.get("b", 0.0))
        c.c = float(p.get("c", 0.0))
    return c


def estimate_mm_per_px(estimated_grating_period_px: Optional[float], grating_pitch_mm: float = 2.0) -> float:
    """force_sensor.estimate_mm_per_px (Code/force_sensor.py:173-187)."""
    if estimated_grating_period_px is None:
        raise RuntimeError("shape_ftp did not return estimated_grating_period_px and OVERRIDE_MM_PER_PX is not set.")
    est = float(estimated_grating_period_px)
    if (not np.isfinite(est)) or est <= 1e-12:
        raise RuntimeError(f"Invalid estimated_grating_period_px={estimated_grating_period_px}.")
    return float(grating_pitch_mm) / est


def predict_force_from_volume(best_model: Dict[str, Any], volume_cm3: float) -> float:
    """force_sensor.predict_force_from_volume (Code/force_sensor.py:149-167) through the C ABI."""
    lib = _lib.load()
    c = _curve(best_model, "force calibration JSON")
    out = ctypes.c_double()
    _lib.check(lib.vistaf_predict_force_from_volume(ctypes.byref(c), float(volume_cm3), ctypes.byref(out)))
    return float(out.value)


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def depth_map_to_volume_cm3(height_map_mm, roi_mask, mm_per_px: float, depth_eps_mm: float = 0.01, device="cuda:0"):
    """force_sensor.depth_map_to_volume_cm3 (Code/force_sensor.py:93-123) on the GPU.

    height_map_mm: [h,w] or [B,h,w] (numpy or torch); roi_mask: same shape bool/uint8 or None for
    isfinite(height).  Returns (volume_cm3, contact_area_mm2, max_depth_mm) floats, or a [B,3] array."""
    lib = _lib.load()
    dev = torch.device(device)
    h_t = torch.as_tensor(np.asarray(height_map_mm, dtype=np.float32) if not torch.is_tensor(height_map_mm) else height_map_mm)
    single = h_t.dim() == 2
    if single:
        h_t = h_t[None]
    h_t = h_t.to(dev, torch.float32).contiguous()
    r_t = None
    if roi_mask is not None:
        r_t = torch.as_tensor(np.asarray(roi_mask) if not torch.is_tensor(roi_mask) else roi_mask)
        if r_t.dim() == 2:
            r_t = r_t[None]
        if r_t.shape != h_t.shape:
            raise ValueError("roi_mask shape does not match height_map_mm")
        r_t = r_t.to(dev).to(torch.uint8).contiguous()
    b, hh, ww = h_t.shape
    out = torch.empty((b, 3), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.vistaf_depth_map_to_volume(h_t.data_ptr(), r_t.data_ptr() if r_t is not None else None, b, hh, ww,
                                                  float(mm_per_px), float(depth_eps_mm), out.data_ptr(), _stream_ptr(dev)))
    res = out.cpu().numpy()
    if single:
        return float(res[0, 0]), float(res[0, 1]), float(res[0, 2])
    return res


class FtpSensor:
    """One FTP session: a reference frame, an ROI circle, constants and the two calibration curves.

    The reference frame is demodulated once on the GPU (Code/shape_ftp.py:1632-1639 does it on every
    call); `predict_batch` then runs Code/shape_ftp.py:1641-2037 + the force tail for B frames.
    """

    def __init__(self, reference, roi_circle: Optional[Tuple[int, int, int]] = None, config: Optional[FtpConfig] = None,
                 height_model: Optional[Dict[str, Any]] = None, use_negated_height: bool = True,
                 force_model: Optional[Dict[str, Any]] = None, max_batch: int = 256, device="cuda:0",
                 frame_shape: Optional[Tuple[int, int]] = None):
        """reference: the session's reference frame, or None for a session that only runs `predict_pairs` (every sample then brings its
        own reference frame; `frame_shape` = (h, w) is needed instead)."""
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        if not torch.cuda.is_available():
            raise RuntimeError("FtpSensor needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device)
        self.config = config or FtpConfig.as_shipped()
        if height_model is None or force_model is None:
            raise KeyError("height_model and force_model (the 'best_model' blocks of the calibration JSONs) are required")
        if reference is None:
            if frame_shape is None:
                raise ValueError("a session without a reference frame needs frame_shape=(h, w)")
            ref = None
            self.h, self.w = int(frame_shape[0]), int(frame_shape[1])
        else:
            ref = self._as_frames(reference)
            if ref.shape[0] != 1:
                raise ValueError("reference must be a single frame")
            self.h, self.w = int(ref.shape[1]), int(ref.shape[2])
        if roi_circle is None:
            roi_circle = (self.w // 2, self.h // 2, min(self.h, self.w) // 2 - 1)
        self.roi_circle = tuple(int(v) for v in roi_circle)
        self.max_batch = int(max_batch)
        self.height_model, self.force_model = height_model, force_model
        cc = self.config.to_c()
        hc, fc = _curve(height_model, "calibration"), _curve(force_model, "force calibration JSON")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_ftp_create(ctypes.byref(cc), self.h, self.w, *self.roi_circle, self.max_batch,
                                                   ctypes.byref(hc), int(bool(use_negated_height)), ctypes.byref(fc),
                                                   ctypes.byref(self._h)))
            if ref is not None:
                fmt = self._format_of(ref)
                _lib.check(self._lib.vistaf_ftp_set_reference(self._h, ref.data_ptr(), fmt, _stream_ptr(self.device)))
        self.reference_info = None
        if ref is not None:
            info = (ctypes.c_double * _lib.NREFINFO)()
            _lib.check(self._lib.vistaf_ftp_get_reference_info(self._h, info))
            self.reference_info = self._info_dict(info, 0)

    @staticmethod
    def _info_dict(info, o):
        return {
            "peak_refined": (info[o + 0], info[o + 1]), "k": (info[o + 2], info[o + 3]), "fft_shape": (int(info[o + 4]), int(info[o + 5])),
            "estimated_grating_period_px": info[o + 6], "mm_per_px": info[o + 7],
        }

    # -- helpers ---------------------------------------------------------------------------------
    def _as_frames(self, frames) -> torch.Tensor:
        t = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames))
        if t.dtype not in (torch.uint8, torch.float16):
            raise ValueError("frames must be uint8 or float16")
        if t.dim() == 2:
            t = t[None]
        elif t.dim() == 3 and t.shape[-1] == 3 and t.shape[0] != 3:
            t = t[None]                    # one HxWx3 frame
        if t.dim() not in (3, 4) or (t.dim() == 4 and t.shape[-1] != 3):
            raise ValueError("frames must be [h,w], [B,h,w], [h,w,3] or [B,h,w,3]")
        return t.to(self.device).contiguous()

    @staticmethod
    def _format_of(t: torch.Tensor) -> int:
        if t.dtype == torch.uint8:
            return _lib.FMT_BGR_U8 if t.dim() == 4 else _lib.FMT_GRAY_U8
        return _lib.FMT_BGR_F16 if t.dim() == 4 else _lib.FMT_GRAY_F16

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.vistaf_ftp_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- batched device API ------------------------------------------------------------------------
    def predict_batch(self, frames, out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """frames: [B,h,w] or [B,h,w,3], uint8 or float16 (device or host).  Returns device tensors:
        height_map_mm [B,h,w] f32 (NaN outside ROI), output_reliable [B,h,w] u8, scalars [B,16] f64, status [B] i32."""
        t = self._as_frames(frames)
        b = int(t.shape[0])
        if t.shape[1] != self.h or t.shape[2] != self.w:
            raise RuntimeError("Reference and deformed images have different sizes.")   # shape_ftp.py:1477-1478
        if b > self.max_batch:
            raise RuntimeError(f"batch {b} exceeds max_batch {self.max_batch}")
        if out is None:
            out = {
                "height_map_mm": torch.empty((b, self.h, self.w), dtype=torch.float32, device=self.device),
                "output_reliable": torch.empty((b, self.h, self.w), dtype=torch.uint8, device=self.device),
                "scalars": torch.empty((b, _lib.NSCALARS), dtype=torch.float64, device=self.device),
                "status": torch.empty((b,), dtype=torch.int32, device=self.device),
            }
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_ftp_predict_batch(
                self._h, t.data_ptr(), self._format_of(t), b, out["height_map_mm"].data_ptr(), out["output_reliable"].data_ptr(),
                out["scalars"].data_ptr(), out["status"].data_ptr(), _stream_ptr(self.device)))
        self._last_out = out
        return out

    def predict_pairs(self, references, frames, out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Uncached pairs: sample b = (references[b], frames[b]); every sample's reference frame is demodulated with its own carrier
        search, as Code/height_to_force.py:384 does by calling shape_ftp.main once per image.  Same outputs as predict_batch; status 3
        marks a sample whose reference spectrum holds no usable carrier."""
        r, t = self._as_frames(references), self._as_frames(frames)
        if r.shape != t.shape or r.dtype != t.dtype:
            raise RuntimeError("Reference and deformed images have different sizes.")   # shape_ftp.py:1477-1478
        b = int(t.shape[0])
        if t.shape[1] != self.h or t.shape[2] != self.w:
            raise RuntimeError("Reference and deformed images have different sizes.")
        if b > self.max_batch:
            raise RuntimeError(f"batch {b} exceeds max_batch {self.max_batch}")
        if out is None:
            out = {
                "height_map_mm": torch.empty((b, self.h, self.w), dtype=torch.float32, device=self.device),
                "output_reliable": torch.empty((b, self.h, self.w), dtype=torch.uint8, device=self.device),
                "scalars": torch.empty((b, _lib.NSCALARS), dtype=torch.float64, device=self.device),
                "status": torch.empty((b,), dtype=torch.int32, device=self.device),
            }
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_ftp_predict_pairs(
                self._h, r.data_ptr(), t.data_ptr(), self._format_of(t), b, out["height_map_mm"].data_ptr(), out["output_reliable"].data_ptr(),
                out["scalars"].data_ptr(), out["status"].data_ptr(), _stream_ptr(self.device)))
        self._last_out = out
        return out

    def pair_info(self, batch: int):
        """reference_info of every sample of the last predict_pairs"""
        info = (ctypes.c_double * (_lib.NREFINFO * batch))()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_ftp_get_pair_info(self._h, batch, info, _stream_ptr(self.device)))
        return [self._info_dict(info, b * _lib.NREFINFO) for b in range(batch)]

    def intermediate(self, name: str, batch: int, dtype=torch.float32) -> torch.Tensor:
        """Copy of a named intermediate plane of the last predict_batch (parity tests)."""
        per = ctypes.c_size_t()
        _lib.check(self._lib.vistaf_ftp_get_intermediate(self._h, name.encode(), None, batch, ctypes.byref(per), None))
        nbytes = per.value * (batch if name not in ("roi", "cref", "amp_ref") else 1)
        buf = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.vistaf_ftp_get_intermediate(self._h, name.encode(), buf.data_ptr(), batch, ctypes.byref(per),
                                                             _stream_ptr(self.device)))
        torch.cuda.synchronize(self.device)
        return buf.view(dtype)

    def _test_set(self, name: str, value: int):
        """Test hook (csrc/test_hooks.h, not part of the boundary): select a fallback / opt-in kernel tier of a stage
        ("inpaint_tier", "flood_tier", "chamfer_twopass", "telea_two_tier", "telea_mw", "unwrap_fast", "big_chain", "fit_capped") or keep debug planes ("keep_planes")."""
        _lib.check(self._lib.vistaf_ftp_test_set(self._h, name.encode(), int(value)))

    def masks(self, index: int = 0) -> Dict[str, np.ndarray]:
        """The seven boolean crop masks of frame `index` of the last predict_batch, under the names the reference stores
        them with in height_map_bundle.npz (Code/shape_ftp.py:1898-1906)."""
        last = getattr(self, "_last_out", None)
        if last is None:
            raise RuntimeError("masks() needs a previous predict_batch")
        batch = int(last["status"].shape[0])

        def plane(name):
            return self.intermediate(name, batch, torch.uint8).view(batch, self.h, self.w)[index].cpu().numpy().astype(bool)
        roi = self.intermediate("roi", 1, torch.uint8).view(self.h, self.w).cpu().numpy().astype(bool)
        cx, cy, r = self.roi_circle
        yy, xx = np.ogrid[:self.h, :self.w]
        circ = ((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r                                   # create_circular_mask (:437-440)
        reliable = plane("reliable")
        # hole candidates exist only without the reliable-region smoothing (Code/shape_ftp.py:1770-1801); otherwise upstream stores zeros too
        holes = plane("hole_cand") if not (self.config.reliable_smooth_sigma_px > 0) else np.zeros((self.h, self.w), dtype=bool)
        return {
            "roi_eroded": roi, "reliable": reliable, "output_reliable": last["output_reliable"][index].cpu().numpy().astype(bool), "circ_mask": circ,
            "contact_kept_by_depth": plane("kept"), "hole_candidates": holes,
            "contact_dilated": plane("contact_d"),
        }

    def enable_stage_timing(self, enable: bool = True):
        _lib.check(self._lib.vistaf_ftp_enable_stage_timing(self._h, int(enable)))

    def stage_times_ms(self) -> Dict[str, float]:
        n = self._lib.vistaf_ftp_stage_count()
        arr = (ctypes.c_float * n)()
        _lib.check(self._lib.vistaf_ftp_get_stage_times(self._h, arr, n))
        return {self._lib.vistaf_ftp_stage_name(i).decode(): float(arr[i]) for i in range(n)}

    # -- single-frame API in the reference's vocabulary ---------------------------------------------
    def predict(self, image) -> Optional[Dict[str, Any]]:
        """One deformed frame -> the dict shape_ftp.main(..., return_results=True) returns
        (Code/shape_ftp.py:2029-2037) plus the force tail of multimodal_sensor.py:388-419.
        Returns None when the reliable mask is empty, as upstream does (shape_ftp.py:1677-1679)."""
        o = self.predict_batch(image)
        torch.cuda.synchronize(self.device)
        status = int(o["status"][0].item())
        if status == _lib.FRAME_EMPTY_RELIABLE:
            return None
        if status != _lib.FRAME_OK:
            raise RuntimeError(f"frame failed with status {status}")
        s = o["scalars"][0].cpu().numpy()
        roi = self.intermediate("roi", 1, torch.uint8).view(self.h, self.w).cpu().numpy().astype(bool)
        res = {
            "height_map_mm_crop": o["height_map_mm"][0].cpu().numpy(),
            "roi_eroded_crop": roi,
            "output_reliable_crop": o["output_reliable"][0].cpu().numpy().astype(bool),
            "estimated_grating_period_px": float(s[5]),
        }
        for i, name in enumerate(SCALAR_NAMES[:-1]):
            if name not in res:
                res[name] = float(s[i])
        res["argmax_depth_index"] = int(s[4])
        res["argmin_unitless_index"] = int(s[8])
        return res


_default_sensor: Optional[FtpSensor] = None


def predict(image, reference=None, **kwargs) -> Optional[Dict[str, Any]]:
    """predict(image[, reference]) -> force map dict.  With `reference` (and, the first time, the keyword
    arguments of FtpSensor) a session is (re)built; later calls reuse it."""
    global _default_sensor
    if reference is not None or _default_sensor is None:
        if reference is None:
            raise RuntimeError("predict() needs a reference frame on first use")
        if _default_sensor is not None:
            _default_sensor.close()
        kwargs.setdefault("max_batch", 1)
        _default_sensor = FtpSensor(reference, **kwargs)
    return _default_sensor.predict(image)
