#!/usr/bin/env python
"""Oracle vs the reference's stored per-image scalars on the 75 force-calibration photographs.

`Force/Height_to_force/calibration_out/per_image_results.csv` (written by Code/height_to_force.py:360-470 with
export_heightmaps=False) holds, for each of `Force/Height_to_force/Loading/*.jpg`, the volume / contact area / max depth the
reference computed against `./Force/FINAL_reference.jpg` -- that path is not in the tree; `Final_demos_images/FINAL_reference.jpg` is the same
photograph (the five demo runs name the same path and reproduce with it, and the stored grating period is identical).  This script runs the alignment oracle + path oracle + tail on every
photograph and writes tests/golden/e2e_loading_report.json (one row per image: stored and recomputed scalars).

    python tests/golden/make_loading_report.py [/root/reference] [workers]
"""
import csv
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
WORKERS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
CIRCLE_PTS = ((1873, 1703), (1599, 707), (2575, 950))


def one(row):
    from oracle import align_oracle as A
    from oracle import ftp_oracle as O
    cfg = O.OracleConfig()
    cal, neg = O.load_calibration(os.path.join(ROOT, "tests", "golden", "calibration_phase_to_height.json"))
    fm = json.load(open(os.path.join(ROOT, "tests", "golden", "calibration_height_to_force.json")))["best_model"]
    rg, dg, (cx, cy, r), info = A.aligned_crops(f"{REF}/Final_demos_images/FINAL_reference.jpg", f"{REF}/Force/Height_to_force/Loading/{row['file']}", CIRCLE_PTS)
    rs = O.make_reference_state(rg, cx, cy, r, cfg)
    out = O.process_frame(dg, rs, cfg, cal, neg, fm)
    rec = {"file": row["file"], "ecc_iters": info["ecc_iters"], "ecc_failed": bool(info["rho"] != info["rho"])}
    for k in ("volume_cm3", "contact_area_mm2", "max_depth_mm", "mm_per_px", "estimated_grating_period_px"):
        rec["stored_" + k] = float(row[k])
        rec[k] = None if out is None else float(out[k])
    return rec


def main():
    rows = list(csv.DictReader(open(f"{REF}/Force/Height_to_force/calibration_out/per_image_results.csv")))
    with ProcessPoolExecutor(WORKERS) as ex:
        recs = list(ex.map(one, rows))
    json.dump(recs, open(os.path.join(ROOT, "tests", "golden", "e2e_loading_report.json"), "w"), indent=1)
    rel = lambda k: np.array([abs(r[k] - r["stored_" + k]) / max(abs(r["stored_" + k]), 1e-12) for r in recs if r[k] is not None])
    for k in ("volume_cm3", "contact_area_mm2", "max_depth_mm"):
        e = rel(k)
        print(k, "median rel", float(np.median(e)), "p90", float(np.percentile(e, 90)), "max", float(e.max()))


if __name__ == "__main__":
    main()
