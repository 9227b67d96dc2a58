"""Generate golden vectors from the REFERENCE's own functions (run in the build container only).

Usage:  python tests/golden/make_golden.py            (needs /root/reference)

The reference (`/root/reference/Code/shape_ftp.py`, `force_sensor.py`) imports `cv2` and
`skimage`, which are not installed here.  Two inert placeholder modules are registered before the
import so that the module body executes; only functions that never touch cv2 are then called
(SURVEY.md §8c lists them).  Outputs are small .npz / .json fixtures committed next to this script;
no reference source or bytecode is copied.

Fixtures written:
  ref_numpy_small.npz   inputs + outputs of the reference's pure-NumPy functions on small seeded arrays
  ref_tail_demos.json   stored result.json scalars of the 5 demo frames + the tail recomputed by the
                        reference's own depth_map_to_volume_cm3 / predict_force_from_volume on the
                        stored height_map_bundle.npz (as multimodal_sensor.py:388-419 calls them)
  ref_tail_demo_E_small.npz  a 4x-decimated stored height map + the reference tail on it
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    for name in ("cv2", "skimage", "skimage.restoration"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage.restoration"].unwrap_phase = None
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, os.path.join(REF, "Code"))
    import shape_ftp
    import force_sensor
    shape_ftp.DEBUG = False
    shape_ftp.DEBUG_LOG_TO_FILE = False
    return shape_ftp, force_sensor


def main():
    S, F = import_reference()
    rng = np.random.default_rng(20261004)
    out = {}

    # --- ROI helpers (shape_ftp.py:383-414)
    out["circle3"] = np.array(S.circle_from_3_points(S.OUTER_CIRCLE_P1, S.OUTER_CIRCLE_P2, S.OUTER_CIRCLE_P3))
    out["mask_64"] = S.create_circular_mask(64, 72, 35, 31, 28)
    out["apo_64"] = S.create_circular_apodization(64, 72, 35, 31, 28, 9)
    out["apo_64_bigtaper"] = S.create_circular_apodization(64, 72, 35, 31, 28, 40)

    # --- FFT peak search (shape_ftp.py:420-503)
    n = 96
    yy, xx = np.mgrid[0:n, 0:n]
    img = (0.3 * np.cos(2 * np.pi * xx / 7.3 + 0.4 * np.sin(yy / 9.0)) + 0.05 * rng.normal(size=(n, n))).astype(np.float32)
    mag = np.abs(np.fft.fftshift(np.fft.fft2(img)))
    out["peaks_mag"] = mag
    peaks = S.find_top_peaks(mag, dc_exclusion=10, n_peaks=12)
    out["peaks_top12"] = np.array(peaks, dtype=np.float64)
    px, py = S.choose_carrier_peak(peaks, n, n)
    out["peaks_chosen"] = np.array([px, py])
    out["peaks_refined"] = np.array(S.refine_peak_parabolic_log(mag, px, py))
    (lx, ly), (lfx, lfy) = S.refine_peak_local_max(mag, px + 2.2, py - 1.4, radius=6)
    out["peaks_local"] = np.array([lx, ly, lfx, lfy], dtype=np.float64)
    out["hann_21"] = S._make_patch_window(21, 21, "hann")

    # --- unwrap (shape_ftp.py:1043-1080): smooth field with wraps, noisy rim, irregular mask
    n = 72
    yy, xx = np.mgrid[0:n, 0:n]
    true = 9.0 * np.exp(-((xx - 30) ** 2 + (yy - 40) ** 2) / (2 * 14.0 ** 2)) + 0.05 * xx
    noise = rng.normal(size=(n, n)) * (0.05 + 1.2 * (np.hypot(xx - 36, yy - 36) > 30))
    wrapped = np.angle(np.exp(1j * (true + noise))).astype(np.float32)
    quality = (np.exp(-((xx - 36) ** 2 + (yy - 36) ** 2) / (2 * 25.0 ** 2)) + 0.02 * rng.random((n, n))).astype(np.float32)
    mask = (np.hypot(xx - 36, yy - 36) <= 33) & ~((np.abs(xx - 50) < 4) & (np.abs(yy - 20) < 6))
    out["uw_wrapped"] = wrapped
    out["uw_quality"] = quality
    out["uw_mask"] = mask
    out["uw_out"] = S.unwrap_quality_guided(wrapped, mask, quality)
    # ties in quality (quantised) exercise the (y, x, py, px) tie-break of the heap tuples
    q2 = np.round(quality * 8) / 8
    out["uw_quality_ties"] = q2.astype(np.float32)
    out["uw_out_ties"] = S.unwrap_quality_guided(wrapped, mask, q2.astype(np.float32))
    out["uw_out_empty"] = S.unwrap_quality_guided(wrapped, np.zeros_like(mask), quality)

    # --- robust polyfit (shape_ftp.py:1086-1136)
    n = 80
    yy, xx = np.mgrid[0:n, 0:n]
    z = (0.8 * (xx / n) - 0.5 * (yy / n) + 0.3 * (xx / n) ** 2 + 0.2 * (xx / n) * (yy / n)
         + 0.02 * rng.normal(size=(n, n))).astype(np.float32)
    z[20:30, 40:55] -= 1.5  # outlier blob (a "contact")
    pm = np.hypot(xx - 40, yy - 38) <= 35
    zz = z.copy(); zz[~pm] = np.nan
    out["pf_z"] = zz
    out["pf_mask"] = pm
    c2, f2 = S.robust_polyfit2d(zz, pm, order=2)
    c1, f1 = S.robust_polyfit2d(zz, pm, order=1)
    out["pf_coef2"], out["pf_fit2"], out["pf_coef1"], out["pf_fit1"] = c2, f2, c1, f1
    cs, fs = S.robust_polyfit2d(zz, pm & (xx < 5) & (yy < 5), order=2)   # < 200 points -> zeros
    out["pf_coef_small"], out["pf_fit_small"] = cs, fs

    # --- small stats helpers, curves (shape_ftp.py:343-354, :682-705, :1277-1284, :1206-1213)
    v = rng.normal(size=1001).astype(np.float32); v[::97] = np.nan
    out["st_v"] = v
    out["st_pcts"] = np.array([S._nanpercentile_safe(v, q) for q in (8.0, 25.0, 92.0, 95, 98, 99.7, 99.9)])
    out["st_median"] = np.array([S._nanmedian_safe(v), S._nanmedian_safe(v[:500])])
    cal, use_neg = S.load_calibration(os.path.join(REF, "Force/Phase_to_height/calibration_out/calibration_model.json"))
    hgrid = np.linspace(-1.5, 0.3, 61).astype(np.float32)
    out["mm_in"] = hgrid
    out["mm_out"] = S.height_unitless_to_depth_mm(hgrid, cal, use_neg)
    out["mm_growth"] = S.model_predict({"type": "growth", "params": {"a": 1.5, "b": 2.0}}, hgrid)
    out["curve_t"] = np.linspace(-0.2, 1.2, 29)
    out["curve_smooth"] = S._curve01(out["curve_t"], "smoothstep")
    cz = rng.normal(size=(8, 9)).astype(np.float32); cz[0, 0] = np.nan
    out["clamp_in"] = cz
    out["clamp_out"] = S.clamp_positive_to_zero(cz, mask=np.ones_like(cz, bool))

    # --- force tail (force_sensor.py:93-187)
    fcal = F.load_force_calibration(os.path.join(REF, "Force/Height_to_force/calibration_out/calibration_model.json"))
    best = fcal["best_model"]
    hm = np.abs(rng.normal(size=(40, 44))).astype(np.float32) * 0.05
    hm[10:20, 12:30] += 0.8
    hm[0:3, :] = np.nan
    roi = np.isfinite(hm)
    out["tail_h"] = hm
    out["tail_res"] = np.array(F.depth_map_to_volume_cm3(hm, roi, 0.0303784, 0.01))
    out["tail_res_neg"] = np.array(F.depth_map_to_volume_cm3(-hm, roi, 0.0303784, 0.01))
    out["tail_res_empty"] = np.array(F.depth_map_to_volume_cm3(hm * 0, roi, 0.0303784, 0.01))
    vols = np.array([0.0, 0.01, 0.1137, 0.2195, 0.5])
    out["force_vols"] = vols
    out["force_growth"] = np.array([F.predict_force_from_volume(best, v_) for v_ in vols])
    for t, p in (("linear0", {"a": 2.0}), ("linear", {"a": 2.0, "b": 0.5}), ("poly2", {"c2": 1.0, "c1": 2.0, "c0": 0.1}),
                 ("sat_exp", {"a": 3.0, "b": 4.0}), ("hinge_saturating", {"a": 3.0, "b": 4.0, "c": 0.05})):
        out[f"force_{t}"] = np.array([F.predict_force_from_volume({"type": t, "params": p}, v_) for v_ in vols])
    out["mm_per_px"] = np.array([F.estimate_mm_per_px(65.83619546657023)])

    np.savez_compressed(os.path.join(HERE, "ref_numpy_small.npz"), **out)

    # --- stored demo goldens: result.json scalars + reference tail recomputed on stored bundles
    demos = {}
    for name in ("FINAL_E_deformed", "FINAL_F_deformed", "FINAL_P_deformed", "FINAL_ROUND_METAL", "FINAL_TEMP_DEMO"):
        d = os.path.join(REF, "Multimodal_Sensor/Demos_report", name, "force_sensing")
        res = json.load(open(os.path.join(d, "result.json")))
        b = np.load(os.path.join(d, "ftp_run", "height_map_bundle.npz"))
        h = b["height_crop"]
        vol, area, md = F.depth_map_to_volume_cm3(h, np.isfinite(h), res["mm_per_px"], res["depth_eps_mm"])
        demos[name] = {
            "stored": {k: res[k] for k in ("estimated_grating_period_px", "mm_per_px", "volume_cm3",
                                           "contact_area_mm2", "max_depth_mm", "force_N", "depth_eps_mm")},
            "reference_tail_on_bundle": {"volume_cm3": vol, "contact_area_mm2": area, "max_depth_mm": md,
                                         "force_N": F.predict_force_from_volume(best, vol)},
            "argmax_depth_index": int(np.nanargmax(h)),
            "reliable_px": int(b["crop_reliable"].sum()),
            "roi_px": int(b["crop_roi_eroded"].sum()),
            "contact_kept_px": int(b["crop_contact_kept_by_depth"].sum()),
            "contact_dilated_px": int(b["crop_contact_dilated"].sum()),
        }
        if name == "FINAL_E_deformed":
            hs = np.ascontiguousarray(h[::4, ::4])
            v2 = F.depth_map_to_volume_cm3(hs, np.isfinite(hs), res["mm_per_px"] * 4, res["depth_eps_mm"])
            np.savez_compressed(os.path.join(HERE, "ref_tail_demo_E_small.npz"), height=hs,
                                tail=np.array(v2), force=np.array([F.predict_force_from_volume(best, v2[0])]),
                                mm_per_px=np.array([res["mm_per_px"] * 4]))
    with open(os.path.join(HERE, "ref_tail_demos.json"), "w") as f:
        json.dump({"force_model": best, "demos": demos}, f, indent=1)
    # calibration constants consumed on the path, copied as data
    for src, dst in (("Force/Phase_to_height/calibration_out/calibration_model.json", "calibration_phase_to_height.json"),
                     ("Force/Height_to_force/calibration_out/calibration_model.json", "calibration_height_to_force.json")):
        with open(os.path.join(REF, src)) as f:
            data = json.load(f)
        with open(os.path.join(HERE, dst), "w") as f:
            json.dump(data, f, indent=1)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
