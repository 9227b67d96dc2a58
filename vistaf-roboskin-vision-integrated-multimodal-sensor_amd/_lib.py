"""ctypes binding of libvistaf_ftp.so (the C ABI declared in include/vistaf_ftp.h).

There is no CPU fallback: if the HIP library is missing or does not load, importing the product
API raises immediately.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvistaf_ftp.so")

NSCALARS = 16
NREFINFO = 8

FMT_GRAY_U8, FMT_BGR_U8, FMT_GRAY_F16, FMT_BGR_F16 = 0, 1, 2, 3
FRAME_OK, FRAME_EMPTY_RELIABLE, FRAME_QUEUE_OVERFLOW, FRAME_NO_CARRIER = 0, 1, 2, 3
CURVE_TYPES = {"linear0": 0, "linear": 1, "poly2": 2, "sat_exp": 3, "growth": 4, "hinge_saturating": 5}

EXPORTS = [
    "vistaf_ftp_abi_version", "vistaf_ftp_last_error", "vistaf_ftp_default_config", "vistaf_ftp_create",
    "vistaf_ftp_set_reference", "vistaf_ftp_get_reference_info", "vistaf_ftp_predict_batch", "vistaf_ftp_predict_pairs",
    "vistaf_ftp_get_pair_info",
    "vistaf_ftp_get_intermediate", "vistaf_ftp_stage_count", "vistaf_ftp_stage_name",
    "vistaf_ftp_enable_stage_timing", "vistaf_ftp_get_stage_times", "vistaf_ftp_destroy",
    "vistaf_depth_map_to_volume", "vistaf_predict_force_from_volume",
]
TEST_EXPORTS = ["vistaf_ftp_test_set"]   # csrc/test_hooks.h: kernel tier selection / debug planes for the parity tests
TEMP_EXPORTS = ["vistaf_tempseg_default_config", "vistaf_tempseg_create", "vistaf_tempseg_destroy", "vistaf_tempseg_segment",
                "vistaf_temp_feature_planes", "vistaf_temp_color_support",
                "vistaf_temp_clamp_map", "vistaf_temp_inpaint_map", "vistaf_temp_fuse_maps", "vistaf_temp_oriented_blur"]   # include/vistaf_temp.h
TEMPSEG_NINFO = 16
ALIGN_EXPORTS = [            # include/vistaf_align.h
    "vistaf_align_default_config", "vistaf_align_create", "vistaf_align_destroy", "vistaf_align_geometry",
    "vistaf_align_set_reference", "vistaf_align_batch",
]


class Curve(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("a", ctypes.c_double), ("b", ctypes.c_double), ("c", ctypes.c_double)]


_INT_FIELDS = [
    "patch_half_width_bins", "dc_exclusion", "fft_pad_px", "roi_erode_px", "apod_taper_px", "reliable_edge_margin_px",
    "poly_order", "frontier_zero_band_px", "valid_close_kernel", "valid_close_iters", "bad_pixel_enable",
    "bad_dilate_ksize", "bad_dilate_iters", "bad_inpaint_radius", "dilate_kernel_size", "dilate_iters", "n_fft_peaks",
    "plane_order_for_removal", "irls_iters", "hole_neighborhood_px", "hole_min_dist_px", "inpaint_radius",
]
_DBL_FIELDS = [
    "pre_blur_sigma_px", "amp_valid_percentile", "quality_smooth_sigma_px", "reliable_smooth_sigma_px", "illum_sigma_px",
    "bad_intensity_percentile", "bad_gradient_percentile", "contact_core_percentile", "contact_percentile",
    "min_contact_frac", "max_contact_frac", "unreliable_smooth_sigma_px", "contact_blob_min_peak_mm",
    "contact_blob_min_peak_rel_frac", "peak_max_dy_from_center", "irls_c", "grating_pitch_mm", "depth_eps_mm", "hole_known_fraction",
]


class CConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in _INT_FIELDS] + [(n, ctypes.c_double) for n in _DBL_FIELDS]


class CTempFuseConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("color_t_min", "color_t_max", "color_guard_band", "switch_margin_c", "final_t_min", "final_t_max")]


class CTempSegConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("seg_band_radius", "seg_dc_exclusion", "seg_illum_sigma", "sat_thresh_gray", "sat_dilate_ksize",
                                                "post_close_kx", "post_close_ky", "post_open_kx", "post_open_ky", "n_peaks")] + \
               [("seg_peak_max_dy_from_center", ctypes.c_double)]


_lib = None


def load():
    """Load libvistaf_ftp.so; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "This package has no CPU path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.vistaf_ftp_abi_version.restype = ci
    lib.vistaf_ftp_last_error.restype = ctypes.c_char_p
    lib.vistaf_ftp_default_config.argtypes = [ctypes.POINTER(CConfig)]
    lib.vistaf_ftp_create.argtypes = [ctypes.POINTER(CConfig), ci, ci, ci, ci, ci, ci, ctypes.POINTER(Curve), ci,
                                      ctypes.POINTER(Curve), ctypes.POINTER(vp)]
    lib.vistaf_ftp_set_reference.argtypes = [vp, vp, ci, vp]
    lib.vistaf_ftp_get_reference_info.argtypes = [vp, ctypes.POINTER(cd)]
    lib.vistaf_ftp_predict_batch.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp]
    lib.vistaf_ftp_predict_pairs.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, vp, vp]
    lib.vistaf_ftp_get_pair_info.argtypes = [vp, ci, ctypes.POINTER(cd), vp]
    lib.vistaf_ftp_get_intermediate.argtypes = [vp, ctypes.c_char_p, vp, ci, ctypes.POINTER(ctypes.c_size_t), vp]
    lib.vistaf_ftp_stage_count.restype = ci
    lib.vistaf_ftp_stage_name.restype = ctypes.c_char_p
    lib.vistaf_ftp_stage_name.argtypes = [ci]
    lib.vistaf_ftp_enable_stage_timing.argtypes = [vp, ci]
    lib.vistaf_ftp_get_stage_times.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ci]
    lib.vistaf_ftp_destroy.argtypes = [vp]
    lib.vistaf_ftp_destroy.restype = None
    lib.vistaf_depth_map_to_volume.argtypes = [vp, vp, ci, ci, ci, cd, cd, vp, vp]
    lib.vistaf_predict_force_from_volume.argtypes = [ctypes.POINTER(Curve), cd, ctypes.POINTER(cd)]
    lib.vistaf_ftp_test_set.argtypes = [vp, ctypes.c_char_p, ci]
    lib.vistaf_tempseg_default_config.argtypes = [ctypes.POINTER(CTempSegConfig)]
    lib.vistaf_tempseg_create.argtypes = [ctypes.POINTER(CTempSegConfig), ci, ci, ctypes.POINTER(vp)]
    lib.vistaf_tempseg_destroy.argtypes = [vp]
    lib.vistaf_tempseg_destroy.restype = None
    lib.vistaf_tempseg_segment.argtypes = [vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(cd), vp]
    lib.vistaf_temp_feature_planes.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp]
    lib.vistaf_temp_color_support.argtypes = [vp, vp, vp, vp, vp, vp, cd, ci, vp, vp, vp]
    lib.vistaf_temp_clamp_map.argtypes = [vp, vp, vp, cd, cd, vp, vp]
    lib.vistaf_temp_inpaint_map.argtypes = [vp, vp, vp, ci, vp, vp]
    lib.vistaf_temp_fuse_maps.argtypes = [vp, vp, vp, vp, ctypes.POINTER(CTempFuseConfig), vp, vp, ctypes.POINTER(ctypes.c_int64), vp]
    lib.vistaf_temp_oriented_blur.argtypes = [vp, vp, vp, cd, cd, cd, vp, vp]
    for fn in EXPORTS + ALIGN_EXPORTS + TEST_EXPORTS + TEMP_EXPORTS:
        getattr(lib, fn)
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().vistaf_ftp_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise RuntimeError(f"vistaf_ftp error {rc}: {msg}")
