#!/usr/bin/env python
"""Oracle vs the stripe-segmentation masks the reference stored for its five demo photographs.

`Multimodal_Sensor/Demos_report/<name>/temperature_sensing/mask_{roi,roi_eff,sat,dark,light,color_support}.png` and `debug_chroma_u8.png` were written by
Code/temperature_sensor.py:803-812 (cropped to the outer-ROI bounding box + 10 px, :770) when Code/multimodal_sensor.py:495-499 ran the
temperature module on `Final_demos_images/<name>.jpg`.  This script runs oracle/temp_oracle.py on the same photographs and writes
tests/golden/temp_seg_report.json (per pair: differing pixels per mask, carrier peak, counts) and, for FINAL_E_deformed (whose photograph
is already a committed fixture), the five stored masks bit-packed into tests/golden/temp_seg_FINAL_E.npz.

    python tests/golden/make_temp_seg_report.py [/root/reference]
"""
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import align_oracle as A          # noqa: E402
from oracle import temp_oracle as T           # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
NAMES = ["FINAL_E_deformed", "FINAL_F_deformed", "FINAL_P_deformed", "FINAL_ROUND_METAL", "FINAL_TEMP_DEMO"]


def main():
    cfg = T.TempSegConfig()
    rows = []
    for name in NAMES:
        t0 = time.time()
        img = A.imread_bgr(f"{REF}/Final_demos_images/{name}.jpg")
        h, w = img.shape[:2]
        roi = T.roi_mask_from_circle(h, w, *cfg.outer_circle)
        y0, y1, x0, x1 = T.bbox_from_mask(roi, cfg.crop_pad_px)
        dark, light, pack = T.segment_dark_light_gratings_periodic_fft(img, roi, cfg)
        planes = T.compute_feature_planes(img, cfg.blur_ksize)
        support, chroma = T.color_support_mask(planes, light, pack["roi_eff"], pack["sat"], cfg)
        got = {"roi": roi, "roi_eff": pack["roi_eff"], "sat": pack["sat"], "dark": dark, "light": light, "color_support": support}
        d = f"{REF}/Multimodal_Sensor/Demos_report/{name}/temperature_sensing"
        stored = {k: np.asarray(Image.open(f"{d}/mask_{k}.png")) > 127 for k in got}
        row = {"name": name, "bbox": [y0, y1, x0, x1], "seconds": round(time.time() - t0, 1)}
        row.update({k: v for k, v in pack["dbg"].items() if k in ("peak_x", "peak_y", "phi0_rad", "chosen", "roi_pixels", "roi_eff_pixels", "sat_pixels",
                                                                    "dark_pixels", "light_pixels", "carrier_period_px")})
        for k in got:
            g = got[k][y0:y1, x0:x1]
            row["shape_equal_" + k] = bool(g.shape == stored[k].shape)
            row["diff_px_" + k] = int((g != stored[k]).sum()) if g.shape == stored[k].shape else -1
            row["stored_px_" + k] = int(stored[k].sum())
        # debug_chroma_u8.png (:820-824): chroma scaled by its 99th percentile inside the ROI
        ch = chroma.copy()
        ch[~roi] = 0
        ch_u8 = np.clip((ch / (np.nanpercentile(ch[roi], 99) + 1e-6)) * 255.0, 0, 255).astype(np.uint8)
        st_ch = np.asarray(Image.open(f"{d}/debug_chroma_u8.png"))
        row["diff_px_chroma_u8"] = int((ch_u8[y0:y1, x0:x1] != st_ch).sum())
        rows.append(row)
        print(json.dumps(row), flush=True)
        if name == "FINAL_E_deformed":
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", "temp_seg_FINAL_E.npz"), bbox=np.array([y0, y1, x0, x1], np.int32),
                                shape=np.array(stored["roi"].shape, np.int32), **{k + "_bits": np.packbits(stored[k]) for k in stored})
    json.dump(rows, open(os.path.join(ROOT, "tests", "golden", "temp_seg_report.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
