"""Result writers with the reference's file schemas, host side only.

* `result_record` / `write_result_json` / `write_result_csv` -- `Code/force_sensor.py:242-295`: the dict written to
  `result.json` (same keys, same order, same nesting of `force_model`) and the one-row `result.csv`.
* `height_map_bundle` / `export_heightmap_files` -- `Code/shape_ftp.py:260-310` and the call at `:1875-1932`:
  `height_map_crop.npy`, `height_map_full.npy`, optional CSVs and `height_map_bundle.npz` with the keys
  `height_crop`, `height_full`, seven `crop_*` and seven `full_*` boolean masks and ten `meta_*` int32 scalars, so the
  bundles the reference stored can be diffed key by key against the ones written from the GPU path.

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
