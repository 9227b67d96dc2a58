"""Force calibration as a GPU batch job: host mirror of `Code/height_to_force.py`.

The reference walks `Force/Height_to_force/Loading/sphere-<k>.jpg` one photograph at a time through `shape_ftp.main`
(:360-470), tabulates volume / area / max depth per image (`per_image_results.csv`), then fits force = f(volume) with six
candidate curves and keeps the one with the smallest RMSE (`calibration_model.json`, :472-515).  Here the photographs go
through `FtpAligner.align` + `FtpSensor.predict_batch` in batches; the curve fitting is the reference's (closed forms,
`numpy.polyfit`, `scipy.optimize.curve_fit` with the same start values and bounds) and stays on the host.
"""
from __future__ import annotations

import csv
import json
import math
import os
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

FORCE_LEVELS_N = (0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 10.0, 15.0, 20.0, 25.0, 30.0, 35.0, 40.0, 45.0)   # height_to_force.py:47
IMAGES_PER_LEVEL = 5                                                                                     # :48
PER_IMAGE_FIELDS = ("file", "force_N", "volume_cm3", "contact_area_mm2", "max_depth_mm", "mm_per_px", "estimated_grating_period_px",
                    "ftp_output_dir")                                                                     # :329-338


def _exp_clip(z):
    return np.exp(np.clip(z, -60.0, 60.0))


# name -> (function, parameter names, equation format): height_to_force.py:115-137
def _hinge(v, a, b, c):
    v = np.asarray(v, float)
    return a * ((1.0 - np.exp(-b * np.maximum(v - c, 0.0))) - (1.0 - np.exp(-b * np.maximum(0.0 - c, 0.0))))


_CURVES: Dict[str, Tuple[Callable, Tuple[str, ...]]] = {
    "linear0": (lambda v, a: a * np.asarray(v, float), ("a",)),
    "linear": (lambda v, a, b: a * np.asarray(v, float) + b, ("a", "b")),
    "poly2": (lambda v, c2, c1, c0: c2 * np.asarray(v, float) ** 2 + c1 * np.asarray(v, float) + c0, ("c2", "c1", "c0")),
    "sat_exp": (lambda v, a, b: a * (1.0 - np.exp(-b * np.maximum(np.asarray(v, float), 0.0))), ("a", "b")),
    "growth": (lambda v, a, b: a * (np.exp(b * np.maximum(np.asarray(v, float), 0.0)) - 1.0), ("a", "b")),
    "hinge_saturating": (_hinge, ("a", "b", "c")),
}
MODEL_CANDIDATES = ("linear0", "linear", "poly2", "sat_exp", "growth", "hinge_saturating")              # :72-79


def _equation(name: str, p: Dict[str, float]) -> str:
    if name == "linear0":
        return f"F = {p['a']:.6g} * V"
    if name == "linear":
        return f"F = {p['a']:.6g} * V + {p['b']:.6g}"
    if name == "poly2":
        return f"F = {p['c2']:.6g} * V^2 + {p['c1']:.6g} * V + {p['c0']:.6g}"
    if name == "sat_exp":
        return f"F = {p['a']:.6g} * (1 - exp(-{p['b']:.6g} * V))"
    if name == "growth":
        return f"F = {p['a']:.6g} * (exp({p['b']:.6g} * V) - 1)"
    return (f"F = {p['a']:.6g} * ( (1-exp(-{p['b']:.6g}*max(V-{p['c']:.6g},0)))"
            f" - (1-exp(-{p['b']:.6g}*max(0-{p['c']:.6g},0))) )")


def model_predict(model: Dict[str, Any], v):
    """height_to_force.predict (:239-255)."""
    name = model["type"]
    if name not in _CURVES:
        raise ValueError(f"Unknown model type: {name}")
    fn, names = _CURVES[name]
    return fn(v, *[float(model["params"][k]) for k in names])


def fit_model(vol_cm3, force_n, name: str) -> Optional[Dict[str, Any]]:
    """height_to_force.fit_model (:139-237): one candidate curve, None when it cannot be fitted."""
    x = np.asarray(vol_cm3, float)
    y = np.asarray(force_n, float)
    fn, names = _CURVES[name]
    if name == "linear0":
        denom = float(np.sum(x * x))
        if denom <= 1e-18:
            return None
        popt = [float(np.sum(x * y) / denom)]
    elif name == "linear":
        coef, *_ = np.linalg.lstsq(np.column_stack([x, np.ones_like(x)]), y, rcond=None)
        popt = [float(coef[0]), float(coef[1])]
    elif name == "poly2":
        if len(x) < 3:
            return None
        popt = [float(v) for v in np.polyfit(x, y, deg=2)]
    else:
        from scipy.optimize import curve_fit
        if name == "hinge_saturating":
            xmax = float(np.max(x)) if len(x) else 1.0
            p0 = [max(np.max(y), 1e-6), 5.0, 0.1 * xmax]
            bounds = ([0.0, 0.0, -0.5 * xmax], [np.inf, np.inf, 1.5 * xmax])
            maxfev = 400000
        else:
            p0 = [max(np.max(y), 1e-6), 1.0]
            bounds = ([0.0, 0.0], [np.inf, np.inf])
            maxfev = 200000
        try:
            popt, _ = curve_fit(fn, x, y, p0=p0, bounds=bounds, maxfev=maxfev)
        except Exception:
            return None
        popt = [float(v) for v in popt]
    params = dict(zip(names, popt))
    return {"type": name, "params": params, "equation": _equation(name, params), "yhat": fn(x, *popt)}


def _rmse(y, yhat) -> float:
    y, yhat = np.asarray(y, float), np.asarray(yhat, float)
    return float(np.sqrt(np.mean((y - yhat) ** 2)))


def _r2(y, yhat) -> float:
    """height_to_force.r2_score (:88-95)."""
    y, yhat = np.asarray(y, float), np.asarray(yhat, float)
    ss_res = float(np.sum((y - yhat) ** 2))
    ss_tot = float(np.sum((y - np.mean(y)) ** 2))
    return float("nan") if ss_tot <= 1e-18 else 1.0 - ss_res / ss_tot


def fit_best_model(x, y):
    """height_to_force.fit_best_model (:257-278): all candidates, best = smallest RMSE, summary sorted by RMSE."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    cands = []
    for name in MODEL_CANDIDATES:
        m = fit_model(x, y, name)
        if m is None:
            continue
        m["sse"] = float(np.sum((y - m["yhat"]) ** 2))
        m["rmse"] = _rmse(y, m["yhat"])
        m["r2"] = _r2(y, m["yhat"])
        cands.append(m)
    if not cands:
        raise RuntimeError("No model could be fit (check your data).")
    best = min(cands, key=lambda d: d["rmse"])
    summary = [{"type": c["type"], "rmse": float(c["rmse"]), "r2": float(c["r2"]), "sse": float(c["sse"])} for c in sorted(cands, key=lambda d: d["rmse"])]
    return best, summary


def calibration_model(rows: Sequence[Dict[str, Any]], reference_path: str, deformed_dir: str, output_dir: str, grating_pitch_mm: float = 2.0,
                      depth_eps_mm: float = 0.01, anchor_origin: bool = True, origin_weight: int = 20) -> Dict[str, Any]:
    """The dict of calibration_model.json (height_to_force.py:472-505) from the per-image rows."""
    if len(rows) < 10:
        raise RuntimeError("Not enough samples processed (check paths / filenames).")
    V = np.array([float(r["volume_cm3"]) for r in rows], float)
    F = np.array([float(r["force_N"]) for r in rows], float)
    if anchor_origin:
        V_fit = np.concatenate([np.zeros(int(origin_weight), float), V])
        F_fit = np.concatenate([np.zeros(int(origin_weight), float), F])
    else:
        V_fit, F_fit = V, F
    best, summary = fit_best_model(V_fit, F_fit)
    return {
        "reference_path": reference_path, "deformed_dir": deformed_dir, "output_dir": output_dir,
        "volume_definition": f"V_cm3 = sum(depth_mm * (mm_per_px^2)) / 1000 over depth>{depth_eps_mm}mm in ROI",
        "grating_pitch_mm": float(grating_pitch_mm), "depth_eps_mm": float(depth_eps_mm),
        "anchor_origin": bool(anchor_origin), "origin_weight": int(origin_weight),
        "best_model": {"type": best["type"], "params": best["params"], "equation": best["equation"], "rmse": float(best["rmse"]),
                       "r2": float(best["r2"]), "sse": float(best["sse"]), "n_fit": int(len(V_fit)), "n_samples": int(len(V))},
        "candidates_summary": summary,
    }


def per_image_rows(aligner, sensor, frames_bgr: Iterable[Tuple[str, np.ndarray]], forces: Sequence[float], batch: int = 8,
                   ftp_output_dir: Callable[[int, str, float], str] = lambda i, f, force: "") -> List[Dict[str, Any]]:
    """The batch job: (file name, decoded BGR photograph) pairs -> the rows of per_image_results.csv.
    `aligner`: FtpAligner built on the reference photograph; `sensor`: FtpSensor built on aligner.reference_gray_crop /
    aligner.circle_crop with the height and force calibrations (the force curve is not used for the table)."""
    import torch
    rows: List[Dict[str, Any]] = []
    pending: List[Tuple[str, np.ndarray]] = []
    items = list(frames_bgr)

    def flush():
        if not pending:
            return
        al = aligner.align(np.stack([p[1] for p in pending]))
        out = sensor.predict_batch(al["aligned_gray"])
        torch.cuda.synchronize()
        sc = out["scalars"].cpu().numpy()
        st = out["status"].cpu().numpy()
        for k, (name, _) in enumerate(pending):
            i = len(rows)
            if int(st[k]) != 0:
                raise RuntimeError(f"{name}: frame status {int(st[k])}")
            period = float(sc[k, 5])
            if not math.isfinite(period) or period <= 1e-9:
                raise RuntimeError(f"{name}: invalid estimated_grating_period_px. Fix shape_ftp return or set OVERRIDE_MM_PER_PX.")
            rows.append({"file": name, "force_N": float(forces[i]), "volume_cm3": float(sc[k, 0]), "contact_area_mm2": float(sc[k, 1]),
                         "max_depth_mm": float(sc[k, 2]), "mm_per_px": float(sc[k, 6]), "estimated_grating_period_px": period,
                         "ftp_output_dir": ftp_output_dir(i, name, float(forces[i]))})
        pending.clear()

    for it in items:
        pending.append(it)
        if len(pending) == batch:
            flush()
    flush()
    return rows


def write_per_image_csv(path: str, rows: Sequence[Dict[str, Any]]) -> str:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", newline="", encoding="utf-8") as f:
        w = csv.DictWriter(f, fieldnames=list(PER_IMAGE_FIELDS))
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in PER_IMAGE_FIELDS})
    return path


def write_calibration_model(path: str, model: Dict[str, Any]) -> str:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(model, f, indent=2)
    return path
