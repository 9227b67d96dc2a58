"""Synthetic fringe frames for benchmarks and parity tests.

I(x,y) = 128 * s(x,y) * [0.55 + 0.35*cos(2*pi*x/P + phi(x,y))] + n,  clipped to u8
  s   = 1 + 0.15*cos(pi*r/R)           slow illumination field
  n   ~ N(0, 2^2)
  P   = 65.836 * n / 1182              grating period (vertical stripes, carrier along +x)
Reference frame: phi = 0.  Deformed frame: phi = -A*exp(-((x-x0)^2+(y-y0)^2)/(2 sigma^2)),
A ~ U(0.2,1.2) rad, (x0,y0) uniform inside 0.5*R of the centre, sigma ~ U(0.08,0.2)*n.
numpy.random.default_rng(seed) with seed = 1000*config + frame index; the reference frame of a
config uses frame index 999.
"""
from __future__ import annotations

import numpy as np

NATIVE_PERIOD_PX = 65.83619546657023   # stored result.json, Multimodal_Sensor/Demos_report/*/force_sensing
NATIVE_CROP = 1182


def roi_circle(n: int):
    """Inscribed ROI disc of an n x n crop: centre (n//2, n//2), radius n//2 - 1."""
    return n // 2, n // 2, n // 2 - 1


def _base(n: int, phi, rng, period=None):
    p = NATIVE_PERIOD_PX * n / NATIVE_CROP if period is None else float(period)
    cx, cy, r = roi_circle(n)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    rr = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
    s = 1.0 + 0.15 * np.cos(np.pi * rr / r)
    img = 128.0 * s * (0.55 + 0.35 * np.cos(2.0 * np.pi * xx / p + phi)) + rng.normal(0.0, 2.0, size=(n, n))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def reference_frame(n: int, config: int = 3, period=None) -> np.ndarray:
    rng = np.random.default_rng(1000 * config + 999)
    return _base(n, 0.0, rng, period)


def deformed_frame(n: int, index: int, config: int = 3, period=None, amp_scale: float = 1.0) -> np.ndarray:
    """amp_scale > 1 multiplies the phase amplitude (e.g. 8 gives genuine 2*pi wraps for unwrap tests)."""
    rng = np.random.default_rng(1000 * config + index)
    cx, cy, r = roi_circle(n)
    amp = rng.uniform(0.2, 1.2) * amp_scale
    rad = 0.5 * r * np.sqrt(rng.uniform(0.0, 1.0))
    ang = rng.uniform(0.0, 2.0 * np.pi)
    x0, y0 = cx + rad * np.cos(ang), cy + rad * np.sin(ang)
    sig = rng.uniform(0.08, 0.2) * n
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    phi = -amp * np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2.0 * sig * sig))
    return _base(n, phi, rng, period)


def deformed_batch(n: int, start: int, count: int, config: int = 3, period=None, amp_scale: float = 1.0) -> np.ndarray:
    return np.stack([deformed_frame(n, start + i, config, period, amp_scale) for i in range(count)], axis=0)
