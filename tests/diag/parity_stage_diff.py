#!/usr/bin/env python3
"""Stage-by-stage GPU vs oracle comparison of ONE synthetic frame (diagnostic; run on the GPU box).
usage: python tests/diag/parity_stage_diff.py <frame_seed> [amp_scale]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vistaf-roboskin-vision-integrated-multimodal-sensor_amd")
from oracle import ftp_oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
seed = int(sys.argv[1])
amp = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = 224
cfg = pkg.FtpConfig.scaled(n)
cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
ref = pkg.synth.reference_frame(n, config=3)
frames = pkg.synth.deformed_batch(n, seed, 1, config=3, amp_scale=amp)
sensor = pkg.FtpSensor(ref, pkg.synth.roi_circle(n), cfg, cal, neg, fm, max_batch=1)
out = sensor.predict_batch(frames)
torch.cuda.synchronize()
rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
o = O.process_frame(frames[0], rs, cfg, cal, neg, fm, keep_intermediates=True)
it = o["inter"]
di = it["demod"]["inter"]


def plane(name, dt=torch.float32):
    return sensor.intermediate(name, 1, dt).cpu().numpy()


def fdiff(name, a, b):
    m = ~(np.isnan(a) | np.isnan(b))
    print("%-12s nan_equal %s  max|diff| %.3e  (scale %.3e)  n>1e-5*scale: %d" % (
        name, np.array_equal(np.isnan(a), np.isnan(b)), float(np.abs(a[m] - b[m]).max()) if m.any() else 0.0, float(np.abs(b[m]).max()) if m.any() else 0.0,
        int((np.abs(a[m] - b[m]) > 1e-5 * np.abs(b[m]).max()).sum()) if m.any() else 0))


def bdiff(name, a, b):
    print("%-12s differing pixels: %d" % (name, int((a != b).sum())))


bdiff("bad", plane("bad1", torch.uint8).reshape(n, n) != 0, di["bad"])
fdiff("inpainted", plane("img").reshape(n, n), di["img_inpainted"])
fdiff("quality", plane("quality").reshape(n, n), it["quality"])
bdiff("rel0", plane("rel0", torch.uint8).reshape(n, n) != 0, it["thresholded"])
bdiff("reliable", plane("reliable", torch.uint8).reshape(n, n) != 0, o["reliable"])
fdiff("unwrapped", plane("unwrapped").reshape(n, n), it["unwrapped"])
bdiff("contact_d", plane("contact_d", torch.uint8).reshape(n, n) != 0, o["contact_dilated"])
bdiff("background", plane("background", torch.uint8).reshape(n, n) != 0, it["background"])
bdiff("kept", plane("kept", torch.uint8).reshape(n, n) != 0, o["contact_kept_by_depth"])
fdiff("unitless", plane("unitless").reshape(n, n), o["height_unitless"])
hm = out["height_map_mm"][0].cpu().numpy()
fdiff("height_mm", hm, o["height_map_mm_crop"])
d = np.abs(hm - o["height_map_mm_crop"])
d[np.isnan(d)] = 0
ys, xs = np.where(d > 0.25 * d.max())
print("worst pixels: rows %d-%d cols %d-%d, value there %.4f vs %.4f" % (ys.min(), ys.max(), xs.min(), xs.max(), hm[ys[0], xs[0]], o["height_map_mm_crop"][ys[0], xs[0]]))
