"""End-to-end pin of the oracle (and of the HIP path) on a REAL photograph pair of the reference.

Fixture `tests/golden/e2e_FINAL_E_deformed.npz` (made by tests/golden/make_e2e_fixture.py): the reference and the aligned
deformed ROI crops of `Final_demos_images/FINAL_reference.jpg` / `FINAL_E_deformed.jpg` (decode, global shift and ECC
alignment by oracle/align_oracle.py, restating Code/shape_ftp.py:1471-1537) and the height map + masks the REFERENCE
ITSELF stored for that pair (`Multimodal_Sensor/Demos_report/FINAL_E_deformed/force_sensing/ftp_run/height_map_bundle.npz`).

This is the test that pins the cv2-dependent stages of the oracle (blur, Sobel, morphology, connected components,
distance transform, Telea inpaint): none of the reference's own files pins them stage by stage, but its stored output
does end to end.  Tolerances are those of a restated alignment (JPEG decoder and float reductions differ from OpenCV's):
measured on FINAL_E 2.9e-5 mm mean / 2.4e-3 mm max against a 1.12 mm peak, masks equal to 1e-5 of the pixels; the table for all
five stored pairs is tests/golden/e2e_bundles_report.json (mean 2.9e-5 .. 3.1e-4 mm, max 2.4e-3 .. 6.9e-3 mm, reliable-mask
IoU >= 0.99996).  tests/golden/e2e_residual_isolation.json shows that this residual is the size of ONE quantisation step of the
pre-path (a 1/32-px step of warpAffine's shift, +-1 grey level of decoder noise), i.e. alignment / decode noise, not the path.

Round 3: ALL FIVE stored bundles are committed fixtures (e2e_<name>.npz; the reference crop is stored once, in the FINAL_E file) and go
through the HIP path at native 1182 x 1182 with as-shipped constants, together with the five stored result.json tails.
"""
import json
import os

import numpy as np
import pytest

from oracle import ftp_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
N = 1182


NAMES = ["FINAL_E_deformed", "FINAL_F_deformed", "FINAL_P_deformed", "FINAL_ROUND_METAL", "FINAL_TEMP_DEMO"]


def _report_row(name):
    return [r for r in json.load(open(os.path.join(G, "e2e_bundles_report.json"))) if r["name"] == name][0]


def _fixture(name="FINAL_E_deformed"):
    z = np.load(os.path.join(G, f"e2e_{name}.npz"))
    unpack = lambda k: np.unpackbits(z[k])[: N * N].reshape(N, N).astype(bool)
    return {
        "ref": np.load(os.path.join(G, "e2e_FINAL_E_deformed.npz"))["ref_gray"], "def": z["def_gray_aligned"], "circle": tuple(int(v) for v in z["circle"]),
        "ecc_failed": bool(z["ecc_failed"]),
        "height": z["height_crop_reference"], "reliable": unpack("reliable_bits"), "contact_dilated": unpack("contact_dilated_bits"),
        "contact_kept": unpack("contact_kept_bits"), "output_reliable": unpack("output_reliable_bits"),
    }


def _iou(a, b):
    return float((a & b).sum()) / max(1, int((a | b).sum()))


def _check_against_reference(height, output_reliable, fx):
    g = fx["height"]
    assert np.array_equal(np.isfinite(height), np.isfinite(g))                 # NaN layout (ROI) identical
    m = np.isfinite(g)
    d = np.abs(height[m] - g[m])
    peak = float(np.nanmax(g))
    assert abs(float(np.nanmax(height)) - peak) <= 5e-4 * peak                 # measured 9.3e-5 (e2e_bundles_report.json)
    assert float(d.mean()) <= 2e-4 and float(d.max()) <= 6e-3                  # mm; measured 2.9e-5 / 2.4e-3
    assert float(np.percentile(d, 99)) <= 2e-3                                 # measured 3.2e-4
    assert _iou(output_reliable.astype(bool), fx["output_reliable"]) >= 0.9999


def test_oracle_reproduces_reference_bundle_on_real_photo():
    fx = _fixture()
    cfg = O.OracleConfig()
    cal, neg = O.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    rs = O.make_reference_state(fx["ref"], *fx["circle"], cfg)
    out = O.process_frame(fx["def"], rs, cfg, cal, neg, None)
    _check_against_reference(out["height_map_mm_crop"], out["output_reliable_crop"], fx)
    assert _iou(out["reliable"].astype(bool), fx["reliable"]) >= 0.9999
    assert _iou(out["contact_dilated"].astype(bool), fx["contact_dilated"]) >= 0.998
    assert _iou(out["contact_kept_by_depth"].astype(bool), fx["contact_kept"]) >= 0.999


def test_report_covers_all_five_stored_pairs():
    rows = json.load(open(os.path.join(G, "e2e_bundles_report.json")))
    assert [r["name"] for r in rows] == ["FINAL_E_deformed", "FINAL_F_deformed", "FINAL_P_deformed", "FINAL_ROUND_METAL", "FINAL_TEMP_DEMO"]
    for r in rows:
        assert r["nan_layout_equal"] and r["iou_reliable"] >= 0.9999
        assert r["abs_diff_mean_mm"] <= 1e-3 and r["abs_diff_max_mm"] <= 2e-2
        assert abs(r["peak_mm_oracle"] - r["peak_mm_reference"]) <= 1e-2 * r["peak_mm_reference"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/Final_demos_images"), reason="reference tree not present")
def test_alignment_restatement_regenerates_fixture():
    """decode + phase correlation + ECC from the JPEGs reproduce the committed aligned crop (deterministic restatement)."""
    from oracle import align_oracle as A
    fx = _fixture()
    R = "/root/reference/Final_demos_images"
    rg, dg, circle, info = A.aligned_crops(f"{R}/FINAL_reference.jpg", f"{R}/FINAL_E_deformed.jpg", ((1873, 1703), (1599, 707), (2575, 950)))
    assert circle == fx["circle"]
    assert np.array_equal(rg, fx["ref"])
    assert np.array_equal(dg, fx["def"])
    assert 10 <= info["ecc_iters"] < 300 and info["rho"] > 0.8


@pytest.mark.gpu
def test_written_bundle_and_result_match_the_stored_ones(pkg, tmp_path):
    """N1 on the real pair: predict -> masks -> height_map_bundle.npz / result.json, diffed key by key against what the
    reference stored for FINAL_E (bundle masks from the fixture, scalars from its result.json)."""
    fx = _fixture()
    cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fcal = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))
    sensor = pkg.FtpSensor(fx["ref"], fx["circle"], pkg.FtpConfig.as_shipped(), cal, neg, fcal["best_model"], max_batch=1)
    res = sensor.predict(fx["def"])
    assert res is not None
    masks = sensor.masks(0)
    cx, cy, r = fx["circle"]
    bundle = pkg.height_map_bundle(res["height_map_mm_crop"], masks, (1421, 538, 2603, 1720), (2160, 3840), (2012, 1129, 591), (cx, cy, r))
    paths = pkg.export_heightmap_files(str(tmp_path), bundle, save_crop_csv=False)
    z = np.load(paths["bundle_npz"])
    assert len(z.files) == 2 + 7 + 7 + 10
    assert z["height_full"].shape == (2160, 3840) and int(z["meta_roi_radius_crop"]) == 590
    for key, ref_mask, thr in (("crop_reliable", fx["reliable"], 0.9999), ("crop_output_reliable", fx["output_reliable"], 0.9999),
                               ("crop_contact_dilated", fx["contact_dilated"], 0.998), ("crop_contact_kept_by_depth", fx["contact_kept"], 0.999)):
        assert _iou(z[key], ref_mask) >= thr, key
    assert not z["crop_hole_candidates"].any()
    assert np.array_equal(z["crop_roi_eroded"], np.isfinite(fx["height"]))             # NaN outside the eroded ROI upstream too
    stored = json.load(open(os.path.join(G, "ref_tail_demos.json")))["demos"]["FINAL_E_deformed"]["stored"]
    rec = pkg.result_record(res, fcal["best_model"], "./Force/FINAL_reference.jpg", "./Final_demos_images/FINAL_E_deformed.jpg", "o", "o/ftp_run")
    assert abs(rec["estimated_grating_period_px"] - stored["estimated_grating_period_px"]) <= 1e-4 * stored["estimated_grating_period_px"]
    assert abs(rec["mm_per_px"] - stored["mm_per_px"]) <= 1e-4 * stored["mm_per_px"]
    assert abs(rec["max_depth_mm"] - stored["max_depth_mm"]) <= 5e-4 * stored["max_depth_mm"]      # measured 2.0e-4 (restated alignment)
    assert abs(rec["contact_area_mm2"] - stored["contact_area_mm2"]) <= 2e-3 * stored["contact_area_mm2"]
    assert abs(rec["volume_cm3"] - stored["volume_cm3"]) <= 2e-3 * stored["volume_cm3"]
    assert abs(rec["force_N"] - stored["force_N"]) <= 5e-3 * stored["force_N"]


def test_loading_set_report_pins_the_oracle_on_75_more_photographs():
    """tests/golden/e2e_loading_report.json (tests/golden/make_loading_report.py): alignment oracle + path oracle + tail on the 75
    force-calibration photographs against the scalars the reference stored for them (per_image_results.csv).  The depth scale
    agrees everywhere; volume / area agree to 1e-3 on most frames and flip on two frames where a secondary blob sits at the
    blob filter's 1/3-of-peak threshold (kept here, dropped there)."""
    rows = json.load(open(os.path.join(G, "e2e_loading_report.json")))
    assert len(rows) == 75 and not any(r["ecc_failed"] for r in rows)
    rel = lambda k: np.array([abs(r[k] - r["stored_" + k]) / abs(r["stored_" + k]) for r in rows])
    assert (rel("mm_per_px") < 1e-5).all() and (rel("estimated_grating_period_px") < 1e-5).all()      # sub-bin carrier refinement: 8e-7
    d = rel("max_depth_mm")
    assert np.median(d) < 1e-4 and d.max() < 3e-2
    v = rel("volume_cm3")
    assert np.median(v) < 1e-3 and (v < 1e-2).sum() >= 70 and (v < 5e-2).sum() >= 74
    a = rel("contact_area_mm2")
    assert np.median(a) < 1e-3 and (a < 1e-2).sum() >= 70


def test_arg_extremum_locations_under_phase_to_height_constants():
    """tests/golden/e2e_phase_to_height_report.json: alignment oracle + path oracle with the constants of the reference's
    offline calibrator (Code/phase_to_height.py: ROI erode 80, frontier band 300, no plane pre-removal) on its four calibration
    photographs, against `Force/Phase_to_height/calibration_out/calibration_results.csv` -- the reference's stored
    arg-extremum ("contact location") goldens.  All four locations are hit exactly (the fourth was one row off, (722, 589) for (722, 588),
    while the oracle's Gaussian rounded its products before adding; with cv's fused multiply-adds the stored pixel comes back and its
    minimum agrees to 2e-5); the minimum values agree to 3.4e-3 relative (a restated alignment).  This data set is reproduced by OpenCV
    3.x's BGR2GRAY coefficients (with 4.x's the fourth pixel is one row off again and its minimum 2.5e-4 away), the force-path data sets
    by 4.x's (oracle/align_oracle.py): the report records the generation it was made with."""
    rows = json.load(open(os.path.join(G, "e2e_phase_to_height_report.json")))
    assert all(r["gray_generation"] == 3 for r in rows)
    assert [r["stored_xy"] for r in rows] == [[703, 514], [607, 524], [729, 537], [722, 588]]
    for r in rows:
        assert r["xy"] == r["stored_xy"]
        assert abs(r["min"] - r["stored_min"]) <= 5e-3 * abs(r["stored_min"])
    assert abs(rows[3]["min"] - rows[3]["stored_min"]) <= 1e-4 * abs(rows[3]["stored_min"])


# ------------------------------------------------------------------------------------------------------------------------------------
# Round 3: every reference-held golden of the path through the HIP kernels
# ------------------------------------------------------------------------------------------------------------------------------------
def _check_pair_against_stored(height, output_reliable, fx, row):
    """`row`: the oracle-vs-stored line of e2e_bundles_report.json for this pair.  The HIP map equals the oracle's to 1e-4 of the peak per
    pixel (asserted separately), so against the reference's STORED map it may deviate by what the oracle chain deviates (restated
    decode + alignment, see e2e_residual_isolation.json) plus that bar -- and by no more than the caps of the five-pair table."""
    g = fx["height"]
    assert np.array_equal(np.isfinite(height), np.isfinite(g))                 # NaN layout (ROI) identical to the stored bundle
    m = np.isfinite(g)
    d = np.abs(height[m] - g[m])
    peak = float(np.nanmax(g))
    bar = 1e-4 * peak
    assert float(d.max()) <= row["abs_diff_max_mm"] + bar and float(d.max()) <= 8e-3
    assert float(d.mean()) <= row["abs_diff_mean_mm"] + bar and float(d.mean()) <= 4e-4
    assert float(np.percentile(d, 99)) <= row["abs_diff_p99_mm"] + bar
    assert abs(float(np.nanmax(height)) - peak) <= abs(row["peak_mm_oracle"] - row["peak_mm_reference"]) + bar
    assert _iou(output_reliable.astype(bool), fx["output_reliable"]) >= 0.9999


@pytest.mark.parametrize("name", NAMES[1:])
def test_fixture_of_every_stored_bundle_is_consistent(name):
    """data check on CPU (the oracle run itself is recorded in e2e_bundles_report.json; FINAL_E runs it live above)"""
    fx = _fixture(name)
    row = _report_row(name)
    assert fx["def"].shape == (N, N) and fx["def"].dtype == np.uint8 and fx["circle"] == (591, 591, 590)
    assert fx["ecc_failed"] == (row["ecc_iters"] == 0) == (name in ("FINAL_ROUND_METAL", "FINAL_TEMP_DEMO"))
    assert abs(float(np.nanmax(fx["height"])) - row["peak_mm_reference"]) < 1e-9
    assert np.array_equal(np.isfinite(fx["height"]), fx["output_reliable"] | np.isfinite(fx["height"]))
    for k in ("volume_cm3", "contact_area_mm2", "max_depth_mm", "force_N"):
        assert row["tail_rel_dev"][k] <= 2e-2, (name, k)                         # the oracle chain vs the reference's result.json


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_reproduces_every_stored_bundle_and_result_json(pkg, name):
    """All five stored `height_map_bundle.npz` + `result.json` of the reference (Multimodal_Sensor/Demos_report/<name>/force_sensing/),
    HIP path at native size with as-shipped constants, incl. the two pairs whose ECC update fails upstream (unaligned crop used,
    shape_ftp.py:576-578).  Three bars per pair:
      (1) against the oracle on the same aligned crops: 1e-4 of the map's peak on EVERY pixel, output_reliable equal, arg-max index exact;
      (2) against the reference's stored map and masks: the oracle chain's own recorded deviation + (1)  (see _check_pair_against_stored);
      (3) the five stored forces / volumes / areas / max depths: what the oracle chain's tail deviates (e2e_bundles_report.json `tail_rel_dev`:
          restated alignment) + 1e-4 relative (force: times the curve's condition number, capped at 4)."""
    import torch
    fx = _fixture(name)
    row = _report_row(name)
    cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fcal = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))
    fm = fcal["best_model"]
    sensor = pkg.FtpSensor(fx["ref"], fx["circle"], pkg.FtpConfig.as_shipped(), cal, neg, fm, max_batch=1)
    res = sensor.predict(fx["def"])
    torch.cuda.synchronize()
    assert res is not None
    hm, rel = res["height_map_mm_crop"], res["output_reliable_crop"].astype(bool)
    # (1) oracle, same inputs
    ocfg = O.OracleConfig()
    rs = O.make_reference_state(fx["ref"], *fx["circle"], ocfg)
    o = O.process_frame(fx["def"], rs, ocfg, cal, neg, fm)
    ref = o["height_map_mm_crop"]
    peak = float(np.nanmax(np.abs(ref)))
    assert np.array_equal(np.isnan(hm), np.isnan(ref))
    assert float(np.nan_to_num(np.abs(hm - ref)).max()) <= 1e-4 * peak
    assert np.array_equal(rel, o["output_reliable_crop"])
    assert int(res["argmax_depth_index"]) == o["argmax_depth_index"]
    # (2) the reference's stored bundle
    _check_pair_against_stored(hm, rel, fx, row)
    masks = sensor.masks(0)
    for key, ref_mask, rk in (("reliable", fx["reliable"], "iou_reliable"), ("contact_dilated", fx["contact_dilated"], "iou_contact_dilated"),
                              ("contact_kept_by_depth", fx["contact_kept"], "iou_contact_kept")):
        assert _iou(np.asarray(masks[key]).astype(bool), ref_mask) >= row[rk] - 1e-4, key
    # (3) the reference's stored result.json
    stored = json.load(open(os.path.join(G, "ref_tail_demos.json")))["demos"][name]["stored"]
    rec = pkg.result_record(res, fm, "./Force/FINAL_reference.jpg", f"./Final_demos_images/{name}.jpg", "o", "o/ftp_run")
    a, b = fm["params"]["a"], fm["params"]["b"]
    v = stored["volume_cm3"]
    kappa = min(4.0, max(1.0, b * v * np.exp(b * v) / (np.exp(b * v) - 1.0)))             # growth curve a (e^{bV} - 1): |V f'(V) / f(V)|
    for k, extra in (("estimated_grating_period_px", 1.0), ("mm_per_px", 1.0), ("max_depth_mm", 1.0), ("contact_area_mm2", 1.0), ("volume_cm3", 1.0), ("force_N", kappa)):
        dev = abs(rec[k] - stored[k]) / abs(stored[k])
        assert dev <= row["tail_rel_dev"][k] + 1e-4 * extra, (name, k, dev, row["tail_rel_dev"][k])
    # the curve evaluation itself is checked tightly on the GPU's own volume (only the volume carries the 1e-4 bar)
    assert abs(rec["force_N"] - a * (np.exp(b * rec["volume_cm3"]) - 1.0)) <= 1e-12 * max(1.0, abs(rec["force_N"]))
