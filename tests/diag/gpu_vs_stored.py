"""Diagnostic (GPU): the HIP path against everything the reference itself stored for this path -- the five height_map_bundle.npz + result.json
of Multimodal_Sensor/Demos_report/* and the four rows of Force/Phase_to_height/calibration_out/calibration_results.csv -- as one table
(tests/golden/e2e_gpu_report.json; the asserting versions are tests/test_e2e_bundles.py and tests/test_align_gpu.py).

    python tests/diag/gpu_vs_stored.py          (on a GPU box, from the repo root)
"""
import csv
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_e2e_bundles as T  # noqa: E402
import test_align_gpu as TA  # noqa: E402
from oracle import ftp_oracle as O  # noqa: E402

pkg = importlib.import_module("vistaf-roboskin-vision-integrated-multimodal-sensor_amd")
G = T.G
cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
out = {"bundles": [], "arg_min_rows": []}
for name in T.NAMES:
    fx = T._fixture(name)
    sensor = pkg.FtpSensor(fx["ref"], fx["circle"], pkg.FtpConfig.as_shipped(), cal, neg, fm, max_batch=1)
    res = sensor.predict(fx["def"])
    torch.cuda.synchronize()
    hm = res["height_map_mm_crop"]
    g = fx["height"]
    m = np.isfinite(g)
    d = np.abs(hm[m] - g[m])
    ocfg = O.OracleConfig()
    o = O.process_frame(fx["def"], O.make_reference_state(fx["ref"], *fx["circle"], ocfg), ocfg, cal, neg, fm)
    ref = o["height_map_mm_crop"]
    stored = json.load(open(os.path.join(G, "ref_tail_demos.json")))["demos"][name]["stored"]
    rec = pkg.result_record(res, fm, "./Force/FINAL_reference.jpg", f"./Final_demos_images/{name}.jpg", "o", "o/ftp_run")
    masks = sensor.masks(0)
    out["bundles"].append({
        "name": name, "ecc_failed_upstream": bool(fx["ecc_failed"]), "unwrap_check_settled": int(sensor.intermediate("unwrap_need", 1, torch.int32).cpu().numpy()[0]) == 0,
        "nan_layout_equal_stored": bool(np.array_equal(np.isfinite(hm), m)), "peak_mm_stored": float(np.nanmax(g)), "peak_mm_gpu": float(np.nanmax(hm)),
        "vs_stored_abs_diff_max_mm": float(d.max()), "vs_stored_abs_diff_mean_mm": float(d.mean()), "vs_stored_abs_diff_p99_mm": float(np.percentile(d, 99)),
        "vs_oracle_max_over_peak": float(np.nan_to_num(np.abs(hm - ref)).max() / np.nanmax(np.abs(ref))),
        "argmax_index_equals_oracle": int(res["argmax_depth_index"]) == o["argmax_depth_index"],
        "iou_reliable_stored": T._iou(np.asarray(masks["reliable"]).astype(bool), fx["reliable"]),
        "iou_output_reliable_stored": T._iou(res["output_reliable_crop"].astype(bool), fx["output_reliable"]),
        "force_N_gpu": rec["force_N"], "force_N_stored": stored["force_N"], "volume_cm3_gpu": rec["volume_cm3"], "volume_cm3_stored": stored["volume_cm3"],
        "max_depth_mm_gpu": rec["max_depth_mm"], "max_depth_mm_stored": stored["max_depth_mm"],
    })
    print(out["bundles"][-1], flush=True)
    del sensor
photos = [TA._imread_bgr(os.path.join(G, "FINAL_reference.jpg"))]
stored = list(csv.DictReader(open(os.path.join(G, "ref_phase_to_height_results.csv"))))
al = pkg.FtpAligner(photos[0], max_batch=4, gray_coeffs=1)
sensor = pkg.FtpSensor(al.reference_gray_crop, al.circle_crop, pkg.FtpConfig.phase_to_height(), cal, neg, fm, max_batch=4)
items = [(r["file"], TA._imread_bgr(os.path.join(G, r["file"]))) for r in stored]
rows = pkg.calibrate.phase_to_height_rows(al, sensor, items, [float(r["depth_mm"]) for r in stored], batch=4)
for r, s in zip(rows, stored):
    out["arg_min_rows"].append({"file": r["file"], "xy_gpu": [r["min_x"], r["min_y"]], "xy_stored": [int(s["min_x"]), int(s["min_y"])],
                                "min_gpu": r["min_height_unitless"], "min_stored": float(s["min_height_unitless"])})
    print(out["arg_min_rows"][-1], flush=True)
json.dump(out, open(os.path.join(G, "e2e_gpu_report.json"), "w"), indent=1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "e2e_gpu_report.json"), "w"), indent=1)
