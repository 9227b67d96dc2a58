"""N2: the GPU pre-path alignment against its CPU restatement (oracle/align_oracle.py) and, end to end,
against the height map the reference stored for the FINAL_E pair.

Inputs: the reference's own demo photographs `tests/golden/FINAL_reference.jpg` / `FINAL_E_deformed.jpg` (data files, decoded
on the host with Pillow as cv2.imread would); expected: the aligned crop of `tests/golden/e2e_FINAL_E_deformed.npz`
(produced by the oracle) and that fixture's stored reference output.

Tolerances: the shift and the ECC warp are float computations (hipFFT vs NumPy FFT, GPU vs NumPy reductions):
shift within 5e-3 px, warp translation within 2e-3 px and rotation within 2e-6 rad of the oracle's; the aligned uint8 crop may
then differ from the oracle's by one grey level on a small fraction of the pixels.
"""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
pytestmark = pytest.mark.gpu


def _imread_bgr(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


@pytest.fixture(scope="module")
def photos():
    return _imread_bgr(os.path.join(G, "FINAL_reference.jpg")), _imread_bgr(os.path.join(G, "FINAL_E_deformed.jpg"))


@pytest.fixture(scope="module")
def fixture():
    return np.load(os.path.join(G, "e2e_FINAL_E_deformed.npz"))


def test_geometry_and_reference_crop(pkg, photos, fixture):
    al = pkg.FtpAligner(photos[0], max_batch=1)
    assert al.circle_full == (2012, 1129, 591)
    assert al.crop_box == (1421, 538, 2603, 1720) and al.crop_shape == (1182, 1182)
    assert al.circle_crop == tuple(int(v) for v in fixture["circle"])
    assert np.array_equal(al.reference_gray_crop.cpu().numpy(), fixture["ref_gray"])           # fixed-point BGR2GRAY: bit-exact


def test_global_shift_only_matches_oracle(pkg, photos):
    """phase correlation + fixed-point warpAffine of the ROI window, ECC off"""
    from oracle import align_oracle as A
    al = pkg.FtpAligner(photos[0], use_ecc=False, max_batch=1)
    out = al.align(photos[1])
    shift_o, resp_o = A.estimate_global_shift(A.bgr2gray_u8(photos[0]).astype(np.float32), A.bgr2gray_u8(photos[1]).astype(np.float32))
    # float32 FFTs (hipFFT vs pocketfft) of a WHITENED cross-power spectrum (every bin has unit magnitude, so rounding noise is not
    # averaged down): the 5x5 centroid moves by up to ~1e-2 px between two float32 transforms (measured 2.2e-3 .. 8.7e-3 as the blur's
    # rounding changed).  warpAffine quantises the shift to 1/32 px, so what matters downstream is the same quantised shift:
    assert abs(out["shift"][0, 0] - shift_o[0]) <= 2e-2 and abs(out["shift"][0, 1] - shift_o[1]) <= 2e-2
    assert np.array_equal(np.rint(np.asarray(out["shift"][0], np.float64) * 32), np.rint(np.asarray(shift_o, np.float64) * 32))
    assert abs(out["response"][0] - resp_o) <= 2e-2 * abs(resp_o)      # sum of a 5x5 window of a noise-dominated whitened correlation: 0.8 % measured
    # same shift -> the warped, cropped, grey-converted window must be the oracle's bit for bit
    M = np.array([[1, 0, np.float32(out["shift"][0, 0])], [0, 1, np.float32(out["shift"][0, 1])]], np.float32)
    x1, y1, x2, y2 = al.crop_box
    exp = A.bgr2gray_u8(A.warp_affine(photos[1], M, False, border="reflect")[y1:y2, x1:x2])
    assert np.array_equal(out["aligned_gray"][0].cpu().numpy(), exp)
    assert not out["ecc_failed"][0] and np.allclose(out["warp"][0], np.eye(2, 3))


def test_ecc_alignment_matches_oracle_and_pins_the_path(pkg, photos, fixture):
    al = pkg.FtpAligner(photos[0], max_batch=1)
    out = al.align(photos[1])
    assert not out["ecc_failed"][0]
    wo = fixture["warp"].astype(np.float64)
    wg = out["warp"][0]
    assert abs(np.arcsin(wg[1, 0]) - np.arcsin(wo[1, 0])) <= 2e-6                 # rotation (rad); the oracle found 5.07e-3
    assert abs(wg[0, 2] - wo[0, 2]) <= 2e-3 and abs(wg[1, 2] - wo[1, 2]) <= 2e-3   # translation (px); (6.53, -4.07)
    assert 10 <= out["ecc_iters"][0] < 300 and abs(out["rho"][0] - 0.86777) <= 1e-4
    got = out["aligned_gray"][0].cpu().numpy()
    exp = fixture["def_gray_aligned"]
    d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
    # two bilinear uint8 warps with 1/32-pixel coordinate quantisation: a 2e-3 px difference in the global shift flips some roundings
    assert d.max() <= 3 and (d > 0).mean() <= 0.10 and (d > 1).mean() <= 2e-3, (int(d.max()), float((d > 0).mean()), float((d > 1).mean()))
    # end to end on the GPU: photographs -> aligned crops -> FTP path, against the height map the reference stored
    import torch
    cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    sensor = pkg.FtpSensor(al.reference_gray_crop, al.circle_crop, pkg.FtpConfig.as_shipped(), cal, neg, fm, max_batch=1)
    o = sensor.predict_batch(out["aligned_gray"])
    torch.cuda.synchronize()
    hm = o["height_map_mm"][0].cpu().numpy()
    g = fixture["height_crop_reference"]
    assert np.array_equal(np.isfinite(hm), np.isfinite(g))
    m = np.isfinite(g)
    dd = np.abs(hm[m] - g[m])
    assert float(dd.mean()) <= 3e-4 and float(dd.max()) <= 8e-3 and abs(float(np.nanmax(hm)) - float(np.nanmax(g))) <= 5e-4 * float(np.nanmax(g))


def test_ecc_failure_returns_the_unaligned_crop(pkg, photos):
    """a blank frame makes the ECC update fail (zero variance): upstream logs the cv2.error and keeps the unaligned crop"""
    al = pkg.FtpAligner(photos[0], max_batch=2)
    blank = np.full_like(photos[1], 90)
    out = al.align(np.stack([photos[1], blank]))
    assert not out["ecc_failed"][0] and out["ecc_failed"][1]
    assert np.isnan(out["rho"][1]) and np.allclose(out["warp"][1], np.eye(2, 3))
    assert (out["aligned_gray"][1].cpu().numpy() == 90).all()


def test_calibration_batch_job_driver(pkg, photos):
    """N4 driver: photographs -> FtpAligner.align -> FtpSensor.predict_batch in batches -> per_image_results rows; every row
    of the (repeated) FINAL_E photograph must equal the single-frame result, and batching must not change it."""
    cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    al = pkg.FtpAligner(photos[0], max_batch=2)
    sensor = pkg.FtpSensor(al.reference_gray_crop, al.circle_crop, pkg.FtpConfig.as_shipped(), cal, neg, fm, max_batch=2)
    items = [(f"sphere-{i + 1}.jpg", photos[1]) for i in range(3)]
    rows = pkg.calibrate.per_image_rows(al, sensor, items, [0.5, 0.5, 1.0], batch=2, ftp_output_dir=lambda i, f, force: f"runs/{i + 1:03d}_{f[:-4]}_F{force}N")
    assert [r["file"] for r in rows] == ["sphere-1.jpg", "sphere-2.jpg", "sphere-3.jpg"] and rows[2]["force_N"] == 1.0
    assert rows[0]["ftp_output_dir"] == "runs/001_sphere-1_F0.5N"
    for k in ("volume_cm3", "contact_area_mm2", "max_depth_mm", "mm_per_px", "estimated_grating_period_px"):
        assert rows[0][k] == rows[1][k] == rows[2][k]                       # same photograph, any batch position: same bits
    stored = {"volume_cm3": 0.11378655442935222, "contact_area_mm2": 304.914771865451, "max_depth_mm": 1.1214957237243652}   # FINAL_E result.json
    for k, tol in (("volume_cm3", 2e-3), ("contact_area_mm2", 2e-3), ("max_depth_mm", 5e-4)):
        assert abs(rows[0][k] - stored[k]) <= tol * stored[k], k


def test_phase_to_height_batch_job_driver(pkg, photos):
    """N4, second calibrator: `calibrate.phase_to_height_rows` (photograph -> FtpAligner -> FtpSensor with the constants of
    Code/phase_to_height.py -> row of calibration_results.csv) on the reference's own calibration photograph
    `Force/Phase_to_height/Height_2mm_deformed.jpg` (tests/golden, a data file).  Expected: the stored row of the reference's
    calibration_results.csv (tests/golden/ref_phase_to_height_results.csv): min_height_unitless -1.26027 at (722, 588); the CPU
    oracle gives -1.26025 at (722, 588) with OpenCV 3.x's BGR2GRAY coefficients -- the generation that reproduces this data set
    (oracle/align_oracle.py; `gray_coeffs = 1` of vistaf_align_config) -- tests/golden/e2e_phase_to_height_report.json.
    Tolerance: 5e-4 unitless (0.04 %) and the stored pixel itself."""
    import csv
    cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    al = pkg.FtpAligner(photos[0], max_batch=2, gray_coeffs=1)
    sensor = pkg.FtpSensor(al.reference_gray_crop, al.circle_crop, pkg.FtpConfig.phase_to_height(), cal, neg, fm, max_batch=2)
    photo = _imread_bgr(os.path.join(G, "Height_2mm_deformed.jpg"))
    rows = pkg.calibrate.phase_to_height_rows(al, sensor, [("Height_2mm_deformed.jpg", photo), ("Height_2mm_deformed.jpg", photo)], [2.07255, 2.07255], batch=2)
    stored = [r for r in csv.DictReader(open(os.path.join(G, "ref_phase_to_height_results.csv"))) if r["file"] == "Height_2mm_deformed.jpg"][0]
    assert rows[0] == rows[1]
    r = rows[0]
    assert r["file"] == stored["file"] and r["depth_mm"] == float(stored["depth_mm"]) and r["heightmap_figure"] == stored["heightmap_figure"]
    assert abs(r["min_height_unitless"] - float(stored["min_height_unitless"])) <= 5e-4, r
    assert (r["min_x"], r["min_y"]) == (int(stored["min_x"]), int(stored["min_y"])), r


def test_all_four_stored_arg_min_locations_through_the_hip_path(pkg, photos):
    """the reference's ONLY stored contact-location goldens -- the four rows of
    `Force/Phase_to_height/calibration_out/calibration_results.csv` (tests/golden/ref_phase_to_height_results.csv: (703,514), (607,524),
    (729,537), (722,588)) -- through `calibrate.phase_to_height_rows` on the reference's own four calibration photographs
    (tests/golden/Height_*mm_deformed.jpg, data files): photograph -> GPU alignment (OpenCV 3.x BGR2GRAY, the generation that
    reproduces this data set) -> HIP path with Code/phase_to_height.py's constants -> (min, x, y).

    Two bars: (a) the HIP path on the ORACLE-aligned crop of each photograph (alignment restatement on the CPU, so that only the path
    differs): the stored pixel EXACTLY and the oracle's minimum to 1e-4; (b) the full GPU chain (GPU alignment + HIP path, one batch of four):
    the stored pixel exactly, the stored minimum within 5e-3 relative (restated alignment; the oracle chain measures 3.4e-3,
    tests/golden/e2e_phase_to_height_report.json)."""
    import csv
    import json
    from oracle import align_oracle as A
    cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    stored = list(csv.DictReader(open(os.path.join(G, "ref_phase_to_height_results.csv"))))
    report = {r["file"]: r for r in json.load(open(os.path.join(G, "e2e_phase_to_height_report.json")))}
    assert [r["file"] for r in stored] == ["Height_0.5mm_deformed.jpg", "Height_1mm_deformed.jpg", "Height_1.5mm_deformed.jpg", "Height_2mm_deformed.jpg"]
    al = pkg.FtpAligner(photos[0], max_batch=4, gray_coeffs=1)
    sensor = pkg.FtpSensor(al.reference_gray_crop, al.circle_crop, pkg.FtpConfig.phase_to_height(), cal, neg, fm, max_batch=4)
    items = [(r["file"], _imread_bgr(os.path.join(G, r["file"]))) for r in stored]
    # (b) the whole chain on the GPU, the four photographs as one batch
    rows = pkg.calibrate.phase_to_height_rows(al, sensor, items, [float(r["depth_mm"]) for r in stored], batch=4)
    for r, s in zip(rows, stored):
        assert (r["min_x"], r["min_y"]) == (int(s["min_x"]), int(s["min_y"])), (r, s)
        assert abs(r["min_height_unitless"] - float(s["min_height_unitless"])) <= 5e-3 * abs(float(s["min_height_unitless"])), (r, s)
        assert r["file"] == s["file"] and r["depth_mm"] == float(s["depth_mm"]) and r["heightmap_figure"] == s["heightmap_figure"]
    # (a) oracle-aligned crops -> HIP path: the path alone
    pts = ((1873, 1703), (1599, 707), (2575, 950))
    i_min, i_arg = pkg.SCALAR_NAMES.index("min_height_unitless"), pkg.SCALAR_NAMES.index("argmin_unitless_index")
    crops = []
    for s in stored:
        rg, dg, circle, _info = A.aligned_crops(os.path.join(G, "FINAL_reference.jpg"), os.path.join(G, s["file"]), pts, gray_generation=3)
        assert circle == al.circle_crop and np.array_equal(rg, al.reference_gray_crop.cpu().numpy())
        crops.append(dg)
    sc = sensor.predict_batch(np.stack(crops))["scalars"].cpu().numpy()
    for j, s in enumerate(stored):
        flat = int(sc[j, i_arg])
        assert (flat % 1182, flat // 1182) == (int(s["min_x"]), int(s["min_y"])), (s["file"], flat % 1182, flat // 1182)
        assert abs(float(sc[j, i_min]) - report[s["file"]]["min"]) <= 1e-4 * abs(report[s["file"]]["min"]), s["file"]
