#!/usr/bin/env python3
"""GPU vs oracle on native-size (1182 x 1182, as-shipped constants) synthetic frames (diagnostic; run on the GPU box).
usage: python tests/diag/native_sweep.py [count] [start]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vistaf-roboskin-vision-integrated-multimodal-sensor_amd")
from oracle import ftp_oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
start = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 1182
cfg = pkg.FtpConfig.as_shipped()
cal, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
ref = pkg.synth.reference_frame(n, config=7)
frames = pkg.synth.deformed_batch(n, start, nb, config=7)
sensor = pkg.FtpSensor(ref, pkg.synth.roi_circle(n), cfg, cal, neg, fm, max_batch=nb)
out = sensor.predict_batch(frames)
torch.cuda.synchronize()
rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
strict = 0
for b in range(nb):
    t0 = time.time()
    o = O.process_frame(frames[b], rs, cfg, cal, neg, fm)
    hm = out["height_map_mm"][b].cpu().numpy()
    rf = o["height_map_mm_crop"]
    nan_eq = bool(np.array_equal(np.isnan(hm), np.isnan(rf)))
    peak = max(float(np.nanmax(np.abs(rf))), 1e-6)
    dev = float(np.nanmax(np.abs(hm - rf))) / peak if nan_eq else float("nan")
    n_diff = int((hm[np.isfinite(rf)] != rf[np.isfinite(rf)]).sum()) if nan_eq else -1
    rel_eq = bool(np.array_equal(out["output_reliable"][b].cpu().numpy().astype(bool), o["output_reliable_crop"]))
    s = out["scalars"][b].cpu().numpy()
    arg_eq = int(s[4]) == int(o["argmax_depth_index"])
    ok = nan_eq and rel_eq and arg_eq and dev <= 1e-4 and int(out["status"][b]) == 0
    strict += ok
    print(f"frame {start + b} {'ok ' if ok else 'BAD'} nan_equal {nan_eq} reliable_equal {rel_eq} argmax_equal {arg_eq} map {dev:.2e} of peak {peak:.4f} mm, {n_diff} differing px ({time.time() - t0:.0f} s oracle)", flush=True)
print(f"strict {strict} / {nb}")
