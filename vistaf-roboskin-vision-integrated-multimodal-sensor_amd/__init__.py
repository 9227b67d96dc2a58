"""MI355X-native (gfx950) implementation of the VISTAF image -> height-map -> force path.

Drop-in for the Fourier-Transform-Profilometry hot path of
rimelq/VISTAF-RoboSkin-Vision-Integrated-Multimodal-Sensor (Code/shape_ftp.py + the force tail of
Code/force_sensor.py).  All image arithmetic runs in hand-written HIP kernels behind the C ABI of
include/vistaf_ftp.h; this package is the thin Python host side.  There is no CPU fallback.
"""
from . import _lib
from .config import FtpConfig
from .ftp import (FtpSensor, SCALAR_NAMES, depth_map_to_volume_cm3, estimate_mm_per_px, load_calibration,
                  load_force_calibration, predict, predict_force_from_volume)
from . import synth
from . import parallel
from .align import FtpAligner, circle_from_3_points
from . import calibrate
from . import tempseg
from .tempseg import TempSegConfig, TempSegmenter, segment_dark_light_gratings_periodic_fft, compute_feature_planes, color_support_mask
from .writers import (export_heightmap_files, height_map_bundle, multimodal_summary, result_record, temperature_statistics,
                      write_multimodal_summary, write_result_csv, write_result_json)

__all__ = ["FtpConfig", "FtpSensor", "SCALAR_NAMES", "depth_map_to_volume_cm3", "estimate_mm_per_px", "load_calibration",
           "load_force_calibration", "predict", "predict_force_from_volume", "synth", "parallel", "FtpAligner", "circle_from_3_points", "calibrate", "_lib", "export_heightmap_files",
           "height_map_bundle", "result_record", "write_result_csv", "write_result_json", "multimodal_summary", "temperature_statistics",
           "write_multimodal_summary", "tempseg", "TempSegConfig", "TempSegmenter", "segment_dark_light_gratings_periodic_fft", "compute_feature_planes", "color_support_mask"]
