"""Result writers with the reference's file schemas, host side only.

* `result_record` / `write_result_json` / `write_result_csv` -- `Code/force_sensor.py:242-295`: the dict written to
  `result.json` (same keys, same order, same nesting of `force_model`) and the one-row `result.csv`.
* `height_map_bundle` / `export_heightmap_files` -- `Code/shape_ftp.py:260-310` and the call at `:1875-1932`:
  `height_map_crop.npy`, `height_map_full.npy`, optional CSVs and `height_map_bundle.npz` with the keys
  `height_crop`, `height_full`, seven `crop_*` and seven `full_*` boolean masks and ten `meta_*` int32 scalars, so the
  bundles the reference stored can be diffed key by key against the ones written from the GPU path.

* `multimodal_summary` / `write_multimodal_summary` -- `Code/multimodal_sensor.py:592-650` with its metric extractors (:214-280): the
  `multimodal_summary.json` of a combined force + temperature session (same keys, order and nesting).

These functions take NumPy arrays (what `FtpSensor.predict` / `FtpSensor.masks` return); nothing here touches the GPU.
"""
from __future__ import annotations

import csv
import json
import math
import os
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple

import numpy as np

CROP_MASK_KEYS = ("roi_eroded", "reliable", "output_reliable", "circ_mask", "contact_kept_by_depth", "hole_candidates",
                  "contact_dilated")                                                    # shape_ftp.py:1898-1906, in that order
RESULT_CSV_FIELDS = ("reference_path", "deformed_path", "volume_cm3", "force_N", "contact_area_mm2", "max_depth_mm", "mm_per_px",
                     "estimated_grating_period_px", "ftp_output_dir", "force_model_type")   # force_sensor.py:269-280


def _safe_float(x, default):
    """force_sensor.safe_float (:60-66): float(x) if finite else default."""
    try:
        v = float(x)
        return v if math.isfinite(v) else default
    except Exception:
        return default


def result_record(res: Mapping[str, Any], best_model: Mapping[str, Any], reference_path: str, deformed_path: str, output_dir: str,
                  ftp_output_dir: str, grating_pitch_mm: float = 2.0, depth_eps_mm: float = 0.01) -> Dict[str, Any]:
    """The dict `force_sensor.main` dumps to result.json (:242-262).  `res`: what `FtpSensor.predict` returned."""
    period = res.get("estimated_grating_period_px", None)
    return {
        "reference_path": reference_path,
        "deformed_path": deformed_path,
        "output_dir": output_dir,
        "ftp_output_dir": ftp_output_dir,
        "grating_pitch_mm": float(grating_pitch_mm),
        "depth_eps_mm": float(depth_eps_mm),
        "estimated_grating_period_px": None if period is None else _safe_float(period, float("nan")),
        "mm_per_px": float(res["mm_per_px"]),
        "volume_cm3": float(res["volume_cm3"]),
        "contact_area_mm2": float(res["contact_area_mm2"]),
        "max_depth_mm": float(res["max_depth_mm"]),
        "force_N": float(res["force_N"]),
        "force_model": {
            "type": best_model.get("type", ""),
            "params": best_model.get("params", {}),
            "equation": best_model.get("equation", ""),
            "rmse": best_model.get("rmse", None),
            "r2": best_model.get("r2", None),
        },
    }


def write_result_json(output_dir: str, record: Mapping[str, Any]) -> str:
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "result.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(record, f, indent=2)
    return path


def write_result_csv(output_dir: str, record: Mapping[str, Any]) -> str:
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "result.csv")
    with open(path, "w", newline="", encoding="utf-8") as f:
        w = csv.DictWriter(f, fieldnames=list(RESULT_CSV_FIELDS))
        w.writeheader()
        row = {k: record[k] for k in RESULT_CSV_FIELDS if k != "force_model_type"}
        row["force_model_type"] = record["force_model"].get("type", "")
        w.writerow(row)
    return path


def height_map_bundle(height_crop: np.ndarray, crop_masks: Mapping[str, np.ndarray], crop_box: Tuple[int, int, int, int],
                      full_shape: Tuple[int, int], circle_full: Tuple[int, int, int], circle_crop: Tuple[int, int, int]) -> Dict[str, np.ndarray]:
    """Arrays of `height_map_bundle.npz` (shape_ftp.py:292-309 with the arguments of :1895-1930).
    crop_box = (x1, y1, x2, y2) of the ROI crop in the full frame; full_shape = (H, W)."""
    x1, y1, x2, y2 = (int(v) for v in crop_box)
    H, W = (int(v) for v in full_shape)
    hc = np.asarray(height_crop).astype(np.float32)
    if hc.shape != (y2 - y1, x2 - x1):
        raise ValueError("height_crop does not match the crop box")
    missing = [k for k in CROP_MASK_KEYS if k not in crop_masks]
    if missing:
        raise ValueError(f"missing crop masks: {missing}")
    full = np.full((H, W), np.nan, np.float32)
    full[y1:y2, x1:x2] = hc
    bundle: Dict[str, np.ndarray] = {"height_crop": hc, "height_full": full}
    for k in CROP_MASK_KEYS:
        bundle[f"crop_{k}"] = np.asarray(crop_masks[k]).astype(bool)
    for k in CROP_MASK_KEYS:
        m = np.zeros((H, W), dtype=bool)
        m[y1:y2, x1:x2] = bundle[f"crop_{k}"]
        bundle[f"full_{k}"] = m
    meta = {
        "crop_x1": x1, "crop_y1": y1, "crop_x2": x2, "crop_y2": y2,
        "roi_center_x_full": circle_full[0], "roi_center_y_full": circle_full[1], "roi_radius_full": circle_full[2],
        "roi_center_x_crop": circle_crop[0], "roi_center_y_crop": circle_crop[1], "roi_radius_crop": circle_crop[2],
    }
    for k, v in meta.items():
        bundle[f"meta_{k}"] = np.asarray(np.int32(v))
    return bundle


def export_heightmap_files(output_dir: str, bundle: Mapping[str, np.ndarray], basename: str = "height_map", save_crop_csv: bool = True,
                           save_full_csv: bool = False) -> Dict[str, str]:
    """shape_ftp.export_heightmap_files (:260-310): <basename>_crop.npy, _full.npy, optional CSVs ("%.9g"), _bundle.npz."""
    os.makedirs(output_dir, exist_ok=True)
    paths = {"crop_npy": os.path.join(output_dir, f"{basename}_crop.npy"), "full_npy": os.path.join(output_dir, f"{basename}_full.npy"),
             "bundle_npz": os.path.join(output_dir, f"{basename}_bundle.npz")}
    np.save(paths["crop_npy"], bundle["height_crop"].astype(np.float32))
    np.save(paths["full_npy"], bundle["height_full"].astype(np.float32))
    if save_crop_csv:
        paths["crop_csv"] = os.path.join(output_dir, f"{basename}_crop.csv")
        np.savetxt(paths["crop_csv"], bundle["height_crop"].astype(np.float32), delimiter=",", fmt="%.9g")
    if save_full_csv:
        paths["full_csv"] = os.path.join(output_dir, f"{basename}_full.csv")
        np.savetxt(paths["full_csv"], bundle["height_full"].astype(np.float32), delimiter=",", fmt="%.9g")
    np.savez_compressed(paths["bundle_npz"], **{k: np.asarray(v) for k, v in bundle.items()})
    return paths


# ---- multimodal_summary.json (Code/multimodal_sensor.py:592-650) ------------------------------------------------------------------
def _nan_float(x) -> float:
    """multimodal_sensor.safe_float (:95-102): float(x) if finite else NaN"""
    return _safe_float(x, float("nan"))


def _phase_to_height_metrics(calib: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
    """extract_phase_to_height_metrics (:214-227)"""
    if calib is None:
        return {}
    best = calib.get("best_model", {})
    return {"calibration_type": "phase_to_height", "model_type": best.get("type", "unknown"), "equation": best.get("equation", ""),
            "r2": _nan_float(best.get("r2", float("nan"))), "rmse": _nan_float(best.get("rmse", float("nan"))),
            "n_samples": int(best.get("n", 0)), "x_definition": calib.get("x_definition", "")}


def _height_to_force_metrics(calib: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
    """extract_height_to_force_metrics (:229-243)"""
    if calib is None:
        return {}
    best = calib.get("best_model", {})
    return {"calibration_type": "height_to_force", "model_type": best.get("type", "unknown"), "equation": best.get("equation", ""),
            "r2": _nan_float(best.get("r2", float("nan"))), "rmse": _nan_float(best.get("rmse", float("nan"))),
            "n_fit": int(best.get("n_fit", 0)), "n_samples": int(best.get("n_samples", 0)),
            "volume_definition": calib.get("volume_definition", "")}


def _temp_model_metrics(calib: Optional[Mapping[str, Any]], model_name: str) -> Dict[str, Any]:
    """extract_temp_model_metrics (:245-280)"""
    if calib is None:
        return {}
    models = calib.get("models_final", {})
    if model_name not in models:
        return {}
    m = models[model_name]

    def block(d):
        return {"rmse_C": _nan_float(d.get("rmse_C", float("nan"))), "mae_C": _nan_float(d.get("mae_C", float("nan"))),
                "r2": _nan_float(d.get("r2", float("nan"))), "max_abs_err_C": _nan_float(d.get("max_abs_err_C", float("nan"))),
                "p95_abs_err_C": _nan_float(d.get("p95_abs_err_C", float("nan"))), "n": int(d.get("n", 0))}
    return {"model": model_name, "degree": int(m.get("degree", 0)), "equation": m.get("equation", ""),
            "frames": block(m.get("metrics_frames", {})), "means": block(m.get("metrics_means", {}))}


def temperature_statistics(temp_map_C, valid_mask) -> Dict[str, Any]:
    """mean / median / std / min / max of the final temperature map over its valid pixels (multimodal_sensor.py:558-567), in the key order
    of the summary's `temperature` block; NaN statistics when no pixel is valid."""
    t = np.asarray(temp_map_C)
    valid = np.asarray(valid_mask, dtype=bool)
    if np.any(valid):
        v = t[valid]
        st = {"mean_C": float(np.mean(v)), "median_C": float(np.median(v)), "std_C": float(np.std(v)), "min_C": float(np.min(v)),
              "max_C": float(np.max(v))}
    else:
        st = {k: float("nan") for k in ("mean_C", "median_C", "std_C", "min_C", "max_C")}
    st["valid_pixels"] = int(np.count_nonzero(valid))
    return st


def multimodal_summary(session_id: str, timestamp: str, reference_image: str, deformed_image: str, session_dir: str, force: Mapping[str, Any],
                       temperature: Mapping[str, Any], p2h_calib: Optional[Mapping[str, Any]], h2f_calib: Optional[Mapping[str, Any]],
                       color_calib: Optional[Mapping[str, Any]], black_calib: Optional[Mapping[str, Any]], force_subdir: str,
                       temp_subdir: str, combined_subdir: str) -> Dict[str, Any]:
    """The dict `multimodal_sensor.main` dumps to combined_outputs/multimodal_summary.json (:592-644).  `force`: what `FtpSensor.predict`
    returned (force_N, volume_cm3, contact_area_mm2, max_depth_mm, mm_per_px); `temperature`: `temperature_statistics(...)`; the four
    calibration dicts are the loaded calibration JSONs (None when a file is missing, as upstream's load_json_safe returns)."""
    return {
        "session_id": session_id,
        "timestamp": timestamp,
        "input_images": {"reference": reference_image, "deformed": deformed_image},
        "output_directory": session_dir,
        "sensor_readings": {
            "force": {"force_N": force["force_N"], "volume_cm3": force["volume_cm3"], "contact_area_mm2": force["contact_area_mm2"],
                      "max_depth_mm": force["max_depth_mm"], "scale_mm_per_px": force["mm_per_px"]},
            "temperature": {k: temperature[k] for k in ("mean_C", "median_C", "std_C", "min_C", "max_C", "valid_pixels")},
        },
        "calibration_performance": {
            "phase_to_height": _phase_to_height_metrics(p2h_calib),
            "height_to_force": _height_to_force_metrics(h2f_calib),
            "temperature_color_model": {k: _temp_model_metrics(color_calib, k) for k in ("heating", "cooling", "global")} if color_calib else {},
            "temperature_black_model": {k: _temp_model_metrics(black_calib, k) for k in ("heating", "cooling", "global")} if black_calib else {},
        },
        "file_paths": {"force_subdir": force_subdir, "temperature_subdir": temp_subdir, "combined_subdir": combined_subdir},
    }


def write_multimodal_summary(combined_subdir: str, summary: Mapping[str, Any]) -> str:
    os.makedirs(combined_subdir, exist_ok=True)
    path = os.path.join(combined_subdir, "multimodal_summary.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(summary, f, indent=2)                      # :648-649
    return path
