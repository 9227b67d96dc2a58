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

from .ftp import SCALAR_NAMES

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


# ---------------------------------------------------------------------------------------------------------------------
# phase_to_height calibration (Code/phase_to_height.py:1264-1383, :1441-1545): unitless height-map minimum -> mm

PHASE_TO_HEIGHT_SAMPLES = (("Height_0.5mm_deformed.jpg", 1.90935), ("Height_1mm_deformed.jpg", 1.94770),
                           ("Height_1.5mm_deformed.jpg", 2.01821), ("Height_2mm_deformed.jpg", 2.07255))      # :36-41
PHASE_TO_HEIGHT_CANDIDATES = ("hinge_saturating", "growth")                                                     # :166
PHASE_TO_HEIGHT_FIELDS = ("file", "depth_mm", "min_height_unitless", "min_x", "min_y", "heightmap_figure")      # :1517-1519


def _p2h_growth(x, a, b):
    return a * (np.exp(b * x) - 1.0)                                                                            # :1264-1266 (no clamp of x)


def _p2h_equation(name: str, p: Dict[str, float]) -> str:
    if name == "growth":
        return f"y = {p['a']:.6g} * (exp({p['b']:.6g} x) - 1)"                                                  # :1299
    return (f"y = {p['a']:.6g} * ((1-exp(-{p['b']:.6g}*max(x-{p['c']:.6g},0)))"
            f" - (1-exp(-{p['b']:.6g}*max(0-{p['c']:.6g},0))))")                                                # :1327-1330


def fit_phase_to_height_model(x, y, name: str, maxfev: int = 200000) -> Optional[Dict[str, Any]]:
    """phase_to_height._fit_growth_curvefit / _fit_hinge_sat_curvefit (:1280-1333): `scipy.optimize.curve_fit` with the
    reference's start values and bounds; None for negative samples or a failed fit."""
    from scipy.optimize import curve_fit
    x, y = np.asarray(x, float), np.asarray(y, float)
    if np.any(x < 0) or np.any(y < 0):
        return None
    if name == "growth":
        fn, names = _p2h_growth, ("a", "b")
        p0 = [max(np.max(y), 1e-6), 1.0]
        bounds = ([0.0, 0.0], [np.inf, np.inf])
    elif name == "hinge_saturating":
        fn, names = _hinge, ("a", "b", "c")
        xmax = float(np.max(x)) if len(x) else 1.0
        p0 = [max(np.max(y), 1e-6), 2.0, 0.2 * xmax]
        bounds = ([0.0, 0.0, -0.5 * xmax], [np.inf, np.inf, 1.2 * xmax])
    else:
        raise ValueError(f"Unknown model type: {name}")
    try:
        popt, _ = curve_fit(fn, x, y, p0=p0, bounds=bounds, maxfev=int(maxfev))
    except Exception:
        return None
    params = {k: float(v) for k, v in zip(names, popt)}
    yhat = fn(x, *[params[k] for k in names])
    return {"type": name, "params": params, "k": len(names), "yhat": yhat, "sse": float(np.sum((y - yhat) ** 2)), "rmse": _rmse(y, yhat),
            "equation": _p2h_equation(name, params)}


def _p2h_r2(y, yhat) -> float:
    """phase_to_height.r2_score (:1064-1071)."""
    y, yhat = np.asarray(y, float), np.asarray(yhat, float)
    ss_res = float(np.sum((y - yhat) ** 2))
    ss_tot = float(np.sum((y - np.mean(y)) ** 2))
    return float("nan") if ss_tot <= 0 else float(1.0 - ss_res / ss_tot)


def fit_phase_to_height_best(x, y, candidates: Sequence[str] = PHASE_TO_HEIGHT_CANDIDATES):
    """phase_to_height.fit_best_model (:1335-1383): best = smallest RMSE; returns (best, summary sorted by RMSE)."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    cands = []
    for name in candidates:
        m = fit_phase_to_height_model(x, y, name)
        if m is None:
            continue
        m["r2"] = _p2h_r2(y, m["yhat"])
        cands.append(m)
    if not cands:
        raise RuntimeError("No valid model candidates could be fit (check your x/y values).")
    best = min(cands, key=lambda d: d["rmse"])
    best["n"] = int(len(x))
    summary = [{"type": c["type"], "rmse": float(c["rmse"]), "r2": float(c["r2"]), "sse": float(c["sse"])} for c in sorted(cands, key=lambda d: d["rmse"])]
    return best, summary


def phase_to_height_model(rows: Sequence[Dict[str, Any]], reference_path: str, deformed_dir: str, output_dir: str, use_negated_height: bool = True,
                          anchor_origin: bool = False, origin_weight: int = 20, apply_origin_correction: bool = False) -> Dict[str, Any]:
    """The dict of Phase_to_height/calibration_out/calibration_model.json (phase_to_height.py:1491-1545) from the per-image rows
    (`file`, `depth_mm`, `min_height_unitless`); rows with a non-finite minimum are skipped as upstream."""
    mins = np.array([float(r["min_height_unitless"]) for r in rows if np.isfinite(float(r["min_height_unitless"]))], float)
    depths = np.array([float(r["depth_mm"]) for r in rows if np.isfinite(float(r["min_height_unitless"]))], float)
    if len(mins) < 2:
        raise RuntimeError("Not enough valid samples to fit a model (need at least 2).")
    x = np.maximum(-mins if use_negated_height else mins, 0.0)
    y = depths
    if anchor_origin:
        w = int(max(1, origin_weight))
        x = np.concatenate((np.zeros(w, float), x))
        y = np.concatenate((np.zeros(w, float), y))
    best, summary = fit_phase_to_height_best(x, y)
    out = {
        "reference_path": reference_path, "deformed_dir": deformed_dir, "output_dir": output_dir,
        "use_negated_height_for_fit": bool(use_negated_height),
        "x_definition": "x = -min_height_unitless" if use_negated_height else "x = min_height_unitless",
        "best_model": {"type": best["type"], "params": best["params"], "equation": best["equation"], "r2": float(best["r2"]),
                       "rmse": float(best["rmse"]), "sse": float(best["sse"]), "n": int(best["n"])},
        "candidates_summary": summary,
        "interpretation": ("This model maps unitless heightmap values to mm. "
                           "If use_negated_height_for_fit=true, it uses x=-height_unitless."),
    }
    if apply_origin_correction:
        fn = _p2h_growth if best["type"] == "growth" else _hinge
        out["best_model"]["origin_correction"] = float(fn(np.array([0.0]), *[best["params"][k] for k in (("a", "b") if best["type"] == "growth" else ("a", "b", "c"))])[0])
    return out


def phase_to_height_rows(aligner, sensor, frames_bgr: Iterable[Tuple[str, np.ndarray]], depths_mm: Sequence[float], batch: int = 4) -> List[Dict[str, Any]]:
    """The per-image loop of phase_to_height.main (:1448-1489) as a batch job: `sensor` must be configured with the constants of
    phase_to_height.py (FtpConfig.phase_to_height()); the row holds the minimum of the unitless height map inside the eroded ROI and
    its location (compute_min_height, :1009-1020: np.nanargmin over the ROI, (x, y) in crop coordinates), which the C ABI returns in the
    scalar record of every frame (`min_height_unitless`, `argmin_unitless_index`)."""
    frames = list(frames_bgr)
    rows: List[Dict[str, Any]] = []
    for i0 in range(0, len(frames), batch):
        chunk = frames[i0:i0 + batch]
        crops = aligner.align(np.stack([f for _, f in chunk]))["aligned_gray"]
        out = sensor.predict_batch(crops)
        sc = out["scalars"].cpu().numpy()
        i_min, i_arg = SCALAR_NAMES.index("min_height_unitless"), SCALAR_NAMES.index("argmin_unitless_index")
        wcrop = int(out["height_map_mm"].shape[-1])
        for j, (name, _f) in enumerate(chunk):
            min_val = float(sc[j, i_min])
            flat = int(sc[j, i_arg])
            mx, my = (flat % wcrop, flat // wcrop) if np.isfinite(min_val) and flat >= 0 else (-1, -1)
            stem = os.path.splitext(name)[0]
            rows.append({"file": name, "depth_mm": float(depths_mm[i0 + j]), "min_height_unitless": min_val, "min_x": mx, "min_y": my,
                         "heightmap_figure": os.path.join(stem, "heightmap.png")})
    return rows


def write_phase_to_height_csv(path: str, rows: Sequence[Dict[str, Any]]) -> str:
    """calibration_results.csv exactly as upstream writes it (:1515-1520: f-string fields, no quoting)."""
    with open(path, "w", encoding="utf-8") as f:
        f.write(",".join(PHASE_TO_HEIGHT_FIELDS) + "\n")
        for r in rows:
            f.write(f"{r['file']},{r['depth_mm']},{r['min_height_unitless']},{r['min_x']},{r['min_y']},{r['heightmap_figure']}\n")
    return path
