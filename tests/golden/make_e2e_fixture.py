#!/usr/bin/env python
"""Builds the real-photograph end-to-end fixtures and prints the oracle-vs-reference table for all five demo pairs.

Inputs (read-only, this container only): the reference's demo photographs `Final_demos_images/FINAL_*.jpg` and the
height-map bundles the reference itself stored for them
(`Multimodal_Sensor/Demos_report/<name>/force_sensing/ftp_run/height_map_bundle.npz`, written by Code/shape_ftp.py:260-310).

For every pair: decode, global shift, ROI crop, ECC alignment (oracle/align_oracle.py, a restatement of
shape_ftp.py:1471-1537), then the path oracle (oracle/ftp_oracle.py) on the aligned crops, compared with the stored
`height_crop` and masks.  For the pairs listed in FIXTURES the aligned inputs and the stored outputs are written to
tests/golden/e2e_<name>.npz (data only: two uint8 crops, the circle, the reference's float32 height map and three of
its masks bit-packed), so that the CPU and GPU parity tests can run the native 1182x1182 path on a real photograph
without the reference tree.

    python tests/golden/make_e2e_fixture.py [/root/reference]
"""
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
NAMES = ["FINAL_E_deformed", "FINAL_F_deformed", "FINAL_P_deformed", "FINAL_ROUND_METAL", "FINAL_TEMP_DEMO"]
FIXTURES = list(NAMES)          # round 3: every stored bundle goes through the HIP path (the reference crop is stored once, in the FINAL_E file)
CIRCLE_PTS = ((1873, 1703), (1599, 707), (2575, 950))      # shape_ftp.py:41-43


def main():
    cfg = O.OracleConfig()
    cal, neg = O.load_calibration(os.path.join(ROOT, "tests", "golden", "calibration_phase_to_height.json"))
    fm = json.load(open(os.path.join(ROOT, "tests", "golden", "calibration_height_to_force.json")))["best_model"]
    stored_tails = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tail_demos.json")))["demos"]     # the reference's five result.json
    rows = []
    ref_crop = None
    for name in NAMES:
        bun = np.load(f"{REF}/Multimodal_Sensor/Demos_report/{name}/force_sensing/ftp_run/height_map_bundle.npz")
        t0 = time.time()
        rg, dg, (cx, cy, r), info = A.aligned_crops(f"{REF}/Final_demos_images/FINAL_reference.jpg", f"{REF}/Final_demos_images/{name}.jpg", CIRCLE_PTS)
        rs = O.make_reference_state(rg, cx, cy, r, cfg)
        out = O.process_frame(dg, rs, cfg, cal, neg, fm)
        hm, g = out["height_map_mm_crop"], bun["height_crop"]
        both = np.isfinite(hm) & np.isfinite(g)
        d = np.abs(hm[both] - g[both])
        iou = lambda a, b: float((a & b).sum()) / max(1, int((a | b).sum()))
        row = {
            "name": name, "shift": [round(float(s), 4) for s in info["shift"]], "ecc_iters": info["ecc_iters"], "rho": (None if info["rho"] != info["rho"] else round(info["rho"], 6)),
            "warp": np.round(info["warp"], 6).tolist(), "nan_layout_equal": bool((np.isfinite(hm) == np.isfinite(g)).all()),
            "peak_mm_oracle": float(np.nanmax(hm)), "peak_mm_reference": float(np.nanmax(g)),
            "abs_diff_max_mm": float(d.max()), "abs_diff_mean_mm": float(d.mean()), "abs_diff_p99_mm": float(np.percentile(d, 99)),
            "iou_reliable": iou(out["reliable"].astype(bool), bun["crop_reliable"]),
            "iou_contact_dilated": iou(out["contact_dilated"].astype(bool), bun["crop_contact_dilated"]),
            "iou_contact_kept": iou(out["contact_kept_by_depth"].astype(bool), bun["crop_contact_kept_by_depth"]),
            "seconds": round(time.time() - t0, 1),
        }
        # the force tail of the oracle chain beside the reference's stored result.json (relative deviations)
        st = stored_tails[name]["stored"]
        row["tail"] = {k: float(out[k]) for k in ("estimated_grating_period_px", "mm_per_px", "volume_cm3", "contact_area_mm2", "max_depth_mm", "force_N")}
        row["tail_rel_dev"] = {k: abs(row["tail"][k] - st[k]) / abs(st[k]) for k in row["tail"]}
        rows.append(row)
        print(json.dumps(row), flush=True)
        if ref_crop is None:
            ref_crop = rg
        assert np.array_equal(rg, ref_crop)          # the reference crop does not depend on the deformed photograph
        if name in FIXTURES:
            extra = {"ref_gray": rg} if name == NAMES[0] else {}
            np.savez_compressed(
                os.path.join(ROOT, "tests", "golden", f"e2e_{name}.npz"),
                def_gray_aligned=dg, ecc_failed=np.array(info["ecc_iters"] == 0), **extra, circle=np.array([cx, cy, r], np.int32),
                height_crop_reference=g.astype(np.float32),
                reliable_bits=np.packbits(bun["crop_reliable"]), contact_dilated_bits=np.packbits(bun["crop_contact_dilated"]),
                contact_kept_bits=np.packbits(bun["crop_contact_kept_by_depth"]), output_reliable_bits=np.packbits(bun["crop_output_reliable"]),
                warp=info["warp"], shift=np.array(info["shift"], np.float64))
    rp = os.path.join(ROOT, "tests", "golden", "e2e_bundles_report.json")
    old = {r["name"]: r for r in json.load(open(rp))} if os.path.exists(rp) else {}
    old.update({r["name"]: r for r in rows})
    json.dump([old[n] for n in NAMES if n in old], open(rp, "w"), indent=1)


if __name__ == "__main__":
    main()
