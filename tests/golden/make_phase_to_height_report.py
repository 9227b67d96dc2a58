#!/usr/bin/env python
"""Oracle vs the arg-extremum goldens the reference's offline calibrator stored.

`Force/Phase_to_height/calibration_out/calibration_results.csv` (written by Code/phase_to_height.py:1459-1520) holds, for four of the
photographs `Force/Phase_to_height/Height_*mm_deformed.jpg`, the minimum unitless height and its pixel (compute_min_height,
Code/phase_to_height.py:1009-1016) -- the reference's only stored "contact location" values.  Its reference photograph is
`./Force/FINAL_reference.jpg` (:23), not in the tree; `Final_demos_images/FINAL_reference.jpg` is the same photograph (see
make_loading_report.py).  This script runs the alignment oracle + the path oracle with that module's constants (ROI erode 80, frontier band
300, no plane pre-removal: Code/phase_to_height.py:63, :115) on those photographs and writes tests/golden/e2e_phase_to_height_report.json.

    python tests/golden/make_phase_to_height_report.py [/root/reference]
"""
import csv
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import align_oracle as A          # noqa: E402
from oracle import ftp_oracle as O            # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
CIRCLE_PTS = ((1873, 1703), (1599, 707), (2575, 950))      # phase_to_height.py uses the same three ROI points as shape_ftp.py:41-43
GRAY_GENERATION = int(os.environ.get("VISTAF_GRAY_GENERATION", "3"))   # this data set is reproduced best by OpenCV 3.x's BGR2GRAY (oracle/align_oracle.py)


def main():
    cfg = O.OracleConfig()
    cfg.roi_erode_px = 80
    cfg.frontier_zero_band_px = 300
    cfg.plane_order_for_removal = 0
    cal, neg = O.load_calibration(os.path.join(ROOT, "tests", "golden", "calibration_phase_to_height.json"))
    rows = list(csv.DictReader(open(f"{REF}/Force/Phase_to_height/calibration_out/calibration_results.csv")))
    out = []
    for row in rows:
        t0 = time.time()
        rg, dg, (cx, cy, r), info = A.aligned_crops(f"{REF}/Final_demos_images/FINAL_reference.jpg", f"{REF}/Force/Phase_to_height/{row['file']}", CIRCLE_PTS, gray_generation=GRAY_GENERATION)
        rs = O.make_reference_state(rg, cx, cy, r, cfg)
        res = O.process_frame(dg, rs, cfg, cal, neg, None)
        v, (x, y) = res["argmin_unitless"]
        rec = {"file": row["file"], "stored_min": float(row["min_height_unitless"]), "stored_xy": [int(row["min_x"]), int(row["min_y"])],
               "min": float(v), "xy": [int(x), int(y)], "ecc_iters": info["ecc_iters"], "gray_generation": GRAY_GENERATION, "s": round(time.time() - t0, 1)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "e2e_phase_to_height_report.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
