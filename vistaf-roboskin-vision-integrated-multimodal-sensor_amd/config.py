"""Constants of the FTP path (reference: Code/shape_ftp.py:23-218, Code/force_sensor.py:33-34)."""
from __future__ import annotations

import dataclasses

from . import _lib

NATIVE_CROP = 1182


@dataclasses.dataclass
class FtpConfig:
    patch_half_width_bins: int = 10
    dc_exclusion: int = 10
    fft_pad_px: int = 96
    roi_erode_px: int = 0
    apod_taper_px: int = 120
    reliable_edge_margin_px: int = 6
    poly_order: int = 2
    frontier_zero_band_px: int = 200
    valid_close_kernel: int = 7
    valid_close_iters: int = 1
    bad_pixel_enable: int = 1
    bad_dilate_ksize: int = 5
    bad_dilate_iters: int = 1
    bad_inpaint_radius: int = 3
    dilate_kernel_size: int = 15
    dilate_iters: int = 2
    n_fft_peaks: int = 12
    plane_order_for_removal: int = 1
    irls_iters: int = 6
    pre_blur_sigma_px: float = 1.5
    amp_valid_percentile: float = 25.0
    quality_smooth_sigma_px: float = 6.0
    reliable_smooth_sigma_px: float = 2.5
    illum_sigma_px: float = 45.0
    bad_intensity_percentile: float = 99.9
    bad_gradient_percentile: float = 99.7
    contact_core_percentile: float = 8.0
    contact_percentile: float = 92.0
    min_contact_frac: float = 0.002
    max_contact_frac: float = 0.40
    unreliable_smooth_sigma_px: float = 9.0
    contact_blob_min_peak_mm: float = 0.1
    contact_blob_min_peak_rel_frac: float = 1.0 / 3.0
    peak_max_dy_from_center: float = 0.12
    irls_c: float = 4.685
    grating_pitch_mm: float = 2.0
    depth_eps_mm: float = 0.01
    # hole stage (Code/shape_ftp.py:140-144): only reached when reliable_smooth_sigma_px == 0 (:1770-1801)
    hole_neighborhood_px: int = 11
    hole_known_fraction: float = 0.70
    hole_min_dist_px: int = 4
    inpaint_radius: int = 5

    @classmethod
    def as_shipped(cls) -> "FtpConfig":
        """The constants exactly as in Code/shape_ftp.py."""
        return cls()

    @classmethod
    def phase_to_height(cls) -> "FtpConfig":
        """The constants of the reference's offline calibrator Code/phase_to_height.py where they differ from shape_ftp.py:
        ROI_ERODE_PX = 80 (:63), FRONTIER_ZERO_BAND_PX = 300 (:115), no debug_ramp plane pre-removal."""
        c = cls()
        c.roi_erode_px = 80
        c.frontier_zero_band_px = 300
        c.plane_order_for_removal = 0
        return c

    @classmethod
    def scaled(cls, n: int) -> "FtpConfig":
        """`scaled-n`: every *_PX constant of shape_ftp.py multiplied by n/1182:
        integer pixel counts rounded to the nearest int >= 1 (0 stays 0), Gaussian sigmas to one decimal
        (>= 0.3), HOLE_NEIGHBORHOOD_PX kept odd >= 3.  Bin counts, kernel sizes and percentiles are unchanged."""
        s = n / NATIVE_CROP
        c = cls()

        def ipx(v):
            return 0 if v == 0 else max(1, int(round(v * s)))

        def sig(v):
            return max(0.3, round(v * s, 1))

        c.fft_pad_px = ipx(c.fft_pad_px)
        c.roi_erode_px = ipx(c.roi_erode_px)
        c.apod_taper_px = ipx(c.apod_taper_px)
        c.reliable_edge_margin_px = ipx(c.reliable_edge_margin_px)
        c.frontier_zero_band_px = ipx(c.frontier_zero_band_px)
        c.hole_min_dist_px = ipx(c.hole_min_dist_px)
        c.hole_neighborhood_px = max(3, ipx(c.hole_neighborhood_px) | 1)
        c.pre_blur_sigma_px = sig(c.pre_blur_sigma_px)
        c.quality_smooth_sigma_px = sig(c.quality_smooth_sigma_px)
        c.reliable_smooth_sigma_px = sig(c.reliable_smooth_sigma_px)
        c.illum_sigma_px = sig(c.illum_sigma_px)
        c.unreliable_smooth_sigma_px = sig(c.unreliable_smooth_sigma_px)
        return c

    def to_c(self) -> "_lib.CConfig":
        cc = _lib.CConfig()
        for name in _lib._INT_FIELDS:
            setattr(cc, name, int(getattr(self, name, 0)))
        for name in _lib._DBL_FIELDS:
            setattr(cc, name, float(getattr(self, name)))
        return cc
