"""CPU-side tests: C-ABI exports, host logic, config, synthetic generator, multi-process sharding (gloo)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_NAME = "vistaf-roboskin-vision-integrated-multimodal-sensor_amd"
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "vistaf_ftp.h")).read()
    declared = sorted(set(re.findall(r"\b(vistaf_\w+)\s*\(", hdr)))
    assert len(declared) >= 14
    lib = pkg._lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(pkg._lib.EXPORTS) == declared
    assert lib.vistaf_ftp_abi_version() == 2
    hdr2 = open(os.path.join(ROOT, "include", "vistaf_align.h")).read()
    declared2 = sorted(set(re.findall(r"\b(vistaf_align_\w+)\s*\(", hdr2)))
    assert len(declared2) == 6
    for name in declared2:
        assert hasattr(lib, name), name
    assert sorted(pkg._lib.ALIGN_EXPORTS) == declared2
    hdr3 = open(os.path.join(ROOT, "include", "vistaf_temp.h")).read()
    declared3 = sorted(set(re.findall(r"\b(vistaf_temp(?:seg)?_\w+)\s*\(", hdr3)))
    assert len(declared3) == 10 and sorted(pkg._lib.TEMP_EXPORTS) == declared3
    for name in declared3:
        assert hasattr(lib, name), name


def test_default_config_is_the_reference_constants(pkg):
    cc = pkg._lib.CConfig()
    assert pkg._lib.load().vistaf_ftp_default_config(ctypes.byref(cc)) == 0
    py = pkg.FtpConfig.as_shipped()
    for name in pkg._lib._INT_FIELDS + pkg._lib._DBL_FIELDS:
        assert getattr(cc, name) == getattr(py, name), name
    # spot values straight from Code/shape_ftp.py:27-218
    assert (cc.fft_pad_px, cc.apod_taper_px, cc.frontier_zero_band_px, cc.illum_sigma_px) == (96, 120, 200, 45.0)
    assert (cc.dilate_kernel_size, cc.dilate_iters, cc.contact_percentile, cc.amp_valid_percentile) == (15, 2, 92.0, 25.0)
    assert (cc.hole_neighborhood_px, cc.hole_min_dist_px, cc.inpaint_radius, cc.hole_known_fraction) == (11, 4, 5, 0.70)   # :140-144
    assert ctypes.sizeof(pkg._lib.CConfig) == 22 * 4 + 19 * 8 and pkg._lib.load().vistaf_ftp_abi_version() == 2


def test_scaled_constants(pkg):
    c = pkg.FtpConfig.scaled(224)
    assert (c.fft_pad_px, c.apod_taper_px, c.frontier_zero_band_px, c.illum_sigma_px) == (18, 23, 38, 8.5)
    assert (c.reliable_edge_margin_px, c.pre_blur_sigma_px, c.quality_smooth_sigma_px) == (1, 0.3, 1.1)
    assert pkg.FtpConfig.scaled(1182).__dict__ == pkg.FtpConfig.as_shipped().__dict__


def test_force_curve_through_c_abi_matches_reference_goldens(pkg):
    g = np.load(os.path.join(G, "ref_numpy_small.npz"))
    best = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    got = np.array([pkg.predict_force_from_volume(best, v) for v in g["force_vols"]])
    assert np.allclose(got, g["force_growth"], rtol=1e-15, atol=0)
    for t, p in (("linear0", {"a": 2.0}), ("linear", {"a": 2.0, "b": 0.5}), ("poly2", {"c2": 1.0, "c1": 2.0, "c0": 0.1}),
                 ("sat_exp", {"a": 3.0, "b": 4.0}), ("hinge_saturating", {"a": 3.0, "b": 4.0, "c": 0.05})):
        got = np.array([pkg.predict_force_from_volume({"type": t, "params": p}, v) for v in g["force_vols"]])
        assert np.allclose(got, g[f"force_{t}"], rtol=1e-15, atol=1e-300), t
    with pytest.raises(ValueError):
        pkg.predict_force_from_volume({"type": "nope", "params": {}}, 0.1)


def test_scale_and_calibration_loaders(pkg, tmp_path):
    g = np.load(os.path.join(G, "ref_numpy_small.npz"))
    assert pkg.estimate_mm_per_px(65.83619546657023) == g["mm_per_px"][0]
    for bad in (None, 0.0, float("nan")):
        with pytest.raises(RuntimeError):
            pkg.estimate_mm_per_px(bad)
    model, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    assert model["type"] == "hinge_saturating" and neg is True
    p = tmp_path / "bad.json"
    p.write_text(json.dumps({"nope": 1}))
    with pytest.raises(ValueError):
        pkg.load_force_calibration(str(p))
    with pytest.raises(KeyError):
        pkg.load_calibration(str(p))


def test_synthetic_generator_is_deterministic(pkg):
    a = pkg.synth.deformed_frame(64, 5)
    b = pkg.synth.deformed_frame(64, 5)
    assert a.dtype == np.uint8 and a.shape == (64, 64) and np.array_equal(a, b)
    assert not np.array_equal(a, pkg.synth.deformed_frame(64, 6))
    assert pkg.synth.roi_circle(224) == (112, 112, 111)
    r = pkg.synth.reference_frame(224)
    spec = np.abs(np.fft.rfft(r[112].astype(float) - r[112].mean()))
    assert abs(np.argmax(spec) - 224 / (65.83619546657023 * 224 / 1182)) <= 1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu(pkg):
    model, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    with pytest.raises(RuntimeError):
        pkg.FtpSensor(pkg.synth.reference_frame(64), None, pkg.FtpConfig.scaled(64), model, neg, fm)


def test_missing_library_raises(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libvistaf_ftp.so")
    with pytest.raises(RuntimeError):
        pkg._lib.load()


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "vistaf-roboskin-vision-integrated-multimodal-sensor_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|import_module\([\"']oracle|oracle/|cvlite", txt, re.M), f


def test_shard_range(pkg):
    par = pkg.parallel if hasattr(pkg, "parallel") else __import__("importlib").import_module(pkg.__name__ + ".parallel")
    assert [par.shard_range(2048, r, 8) for r in range(8)] == [(256 * r, 256 * r + 256) for r in range(8)]
    spans = [par.shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        par.shard_range(10, 4, 4)


def _gloo_worker(rank, world, port, pkg_name, q):
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module(pkg_name + ".parallel")
    total, h = 6, 5
    a, b = par.shard_range(total, rank, world)
    full_h = torch.arange(total * h * h, dtype=torch.float32).reshape(total, h, h)
    full_s = torch.arange(total * 16, dtype=torch.float64).reshape(total, 16)
    local = {"height_map_mm": full_h[a:b].clone(), "scalars": full_s[a:b].clone()}
    got = par.all_gather_outputs(local)
    ok = torch.equal(got["height_map_mm"], full_h) and torch.equal(got["scalars"], full_s)
    # the single-collective variant: maps (f32), scalars (f64), status (i32) and a byte mask packed into one record per frame
    full_st = torch.arange(total, dtype=torch.int32)
    full_m = (torch.arange(total * h * h) % 3 == 0).to(torch.uint8).reshape(total, h, h)
    local2 = dict(local, status=full_st[a:b].clone(), output_reliable=full_m[a:b].clone())
    pg = par.PackedGather(local2, keys=("height_map_mm", "scalars", "status", "output_reliable"))
    for _ in range(2):                                                     # buffers are reused from step to step
        g2 = pg.gather(local2)
    ok = ok and torch.equal(g2["height_map_mm"], full_h) and torch.equal(g2["scalars"], full_s)
    ok = ok and torch.equal(g2["status"], full_st) and torch.equal(g2["output_reliable"], full_m)
    ok = ok and pg.frame_bytes % 8 == 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_all_gather_gloo(pkg):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, pkg.__name__, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _ragged_worker(rank, world, port, pkg_name, q):
    """shard_range(2050, r, 8) gives shards of 257 and 256 frames: the same raggedness with two ranks is total = 515 -> 258 + 257."""
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module(pkg_name + ".parallel")
    sizes8 = [b - a for a, b in (par.shard_range(2050, r, 8) for r in range(8))]
    ok = sizes8 == [257, 257, 256, 256, 256, 256, 256, 256]
    total, h = 515, 6
    a, b = par.shard_range(total, rank, world)
    ok = ok and (b - a) == (258 if rank == 0 else 257)
    full_h = torch.arange(total * h * h, dtype=torch.float32).reshape(total, h, h) + 1.0
    full_s = torch.arange(total * 16, dtype=torch.float64).reshape(total, 16) + 1.0
    full_st = torch.arange(total, dtype=torch.int32) + 1
    local = {"height_map_mm": full_h[a:b].clone(), "scalars": full_s[a:b].clone(), "status": full_st[a:b].clone()}
    # (1) unequal shards without `total`: refused on EVERY rank at construction, before any data-path collective
    try:
        par.PackedGather(local, keys=("height_map_mm", "scalars", "status"))
        raised = False
    except ValueError as e:
        raised = "shards differ" in str(e)
    ok = ok and raised
    # (2) with `total`: the short shard is padded, still ONE collective per step, compact() restores the global order
    calls = {"n": 0}
    orig = dist.all_gather_into_tensor

    def counting(*args, **kw):
        calls["n"] += 1
        return orig(*args, **kw)
    dist.all_gather_into_tensor = counting
    pg = par.PackedGather(local, keys=("height_map_mm", "scalars", "status"), total=total)
    for _ in range(2):
        pg.gather(local)
    dist.all_gather_into_tensor = orig
    c = pg.compact()
    ok = ok and calls["n"] == 2 and pg.padded and pg.b == 258 and pg.sizes == [258, 257]
    ok = ok and torch.equal(c["height_map_mm"], full_h) and torch.equal(c["scalars"], full_s) and torch.equal(c["status"], full_st)
    v = pg.views()
    ok = ok and v["status"].shape[0] == 2 * 258 and int(v["status"][2 * 258 - 1]) == 0          # the padding row of the short shard stays zero
    # (3) a rank that holds the wrong number of frames for `total` is refused
    try:
        par.PackedGather({k: t[:-1] for k, t in local.items()}, keys=("height_map_mm",), total=total)
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_short_last_shard_is_padded_or_refused_gloo(pkg):
    """PackedGather used to assume equal shards silently."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, pkg.__name__, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _bench_flow_worker(rank, world, port, q):
    """bench.py's own step / timing control flow (Stepper, timed_steps, PackedGather with GATHER_KEYS) on CPU tensors of the REAL output
    shapes and dtypes of a 256-frame shard, gloo instead of RCCL; every collective is counted."""
    import contextlib
    import importlib
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench = importlib.import_module("bench")
    par = importlib.import_module(PKG_NAME + ".parallel")
    B, n = 256, 224
    calls = {"gather": 0, "other": 0}
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

    def counting_ag(*a, **k):
        calls["gather"] += 1
        return real_ag(*a, **k)

    def counting_ar(*a, **k):
        calls["other"] += 1
        return real_ar(*a, **k)
    dist.all_gather_into_tensor, dist.all_reduce = counting_ag, counting_ar

    def make_out():
        return {"height_map_mm": torch.empty((B, n, n), dtype=torch.float32), "output_reliable": torch.empty((B, n, n), dtype=torch.uint8),
                "scalars": torch.empty((B, 16), dtype=torch.float64), "status": torch.empty((B,), dtype=torch.int32)}

    class FakeSession:                      # stands in for FtpSensor.predict_batch: rank- and step-dependent content, no GPU
        def __init__(self):
            self.calls = 0

    def run(s_, o_):
        s_.calls += 1
        o_["height_map_mm"].fill_(float(1000 * rank + s_.calls))
        o_["height_map_mm"][:, 0, 0] = torch.arange(B, dtype=torch.float32) + B * rank
        o_["scalars"].copy_((torch.arange(B * 16, dtype=torch.float64) + 1e6 * rank).reshape(B, 16))
        o_["status"].fill_(rank)
        return o_
    sessions, outs = [FakeSession() for _ in range(3)], [make_out() for _ in range(3)]
    gathers = [par.PackedGather(o, keys=bench.GATHER_KEYS) for o in outs]
    stepper = bench.Stepper(run, sessions, outs, [None] * 3, gathers, contextlib.nullcontext)
    warm, steps = 1, 4
    elapsed = bench.timed_steps(stepper, warm, steps, dist, lambda: None, torch.device("cpu"))
    ok = calls["gather"] == warm + steps                                   # exactly ONE data-path collective per step
    ok = ok and calls["other"] == 1                                        # + the max-over-ranks of the elapsed time, once
    ok = ok and stepper.n == warm + steps and elapsed > 0
    g = gathers[(warm + steps - 1) % 3].views()                            # the last step's gathered records
    ok = ok and tuple(g["height_map_mm"].shape) == (world * B, n, n) and g["scalars"].dtype == torch.float64 and g["status"].dtype == torch.int32
    for r in range(world):
        ok = ok and bool((g["status"][r * B:(r + 1) * B] == r).all())
        ok = ok and torch.equal(g["height_map_mm"][r * B:(r + 1) * B, 0, 0], torch.arange(B, dtype=torch.float32) + B * r)
        ok = ok and float(g["scalars"][r * B, 1]) == 1e6 * r + 1
    q.put((rank, bool(ok), calls["gather"], calls["other"]))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_flow_two_ranks_gloo(pkg):
    """N > 1 readiness without hardware: the benchmark's step loop with the real shard shapes ([256,224,224] f32 + [256,16] f64 + [256] i32),
    two gloo ranks on CPU, one collective per step; and the shard ranges of BASELINE configs[3] (2048 / 8) and configs[4] (1024 / 8)."""
    import torch.multiprocessing as mp
    par = pkg.parallel
    for total in (2048, 1024):
        spans = [par.shard_range(total, r, 8) for r in range(8)]
        assert spans[0][0] == 0 and spans[-1][1] == total and all(spans[i][1] == spans[i + 1][0] for i in range(7))
        assert {b - a for a, b in spans} == {total // 8}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_flow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, 5, 1), (1, True, 5, 1)], res


def test_result_writers_match_reference_schema(pkg, tmp_path):
    """N1: result.json / result.csv / height_map_bundle.npz carry the reference's keys, order and dtypes
    (Code/force_sensor.py:242-295, Code/shape_ftp.py:292-309); checked against the stored demo artefacts' schema recorded in
    tests/golden/ref_tail_demos.json and against the bundle key list of SURVEY.md §3."""
    import csv
    n = 32
    rng = np.random.default_rng(0)
    height = rng.uniform(0, 1, (n, n)).astype(np.float32)
    height[:4] = np.nan
    res = {"estimated_grating_period_px": 65.83619546657023, "mm_per_px": 0.030378426119953176, "volume_cm3": 0.11378655442935222,
           "contact_area_mm2": 304.914771865451, "max_depth_mm": 1.1214957237243652, "force_N": 3.2960528395288056}
    fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    rec = pkg.result_record(res, fm, "./Force/FINAL_reference.jpg", "./Final_demos_images/FINAL_E_deformed.jpg", "out", "out/ftp_run")
    assert list(rec.keys()) == ["reference_path", "deformed_path", "output_dir", "ftp_output_dir", "grating_pitch_mm", "depth_eps_mm",
                                "estimated_grating_period_px", "mm_per_px", "volume_cm3", "contact_area_mm2", "max_depth_mm", "force_N",
                                "force_model"]
    assert list(rec["force_model"].keys()) == ["type", "params", "equation", "rmse", "r2"]
    jp = pkg.write_result_json(str(tmp_path), rec)
    back = json.load(open(jp))
    assert back == json.loads(json.dumps(rec)) and open(jp).read().startswith('{\n  "reference_path"')       # indent=2 as upstream
    # the stored FINAL_E result.json values survive the round trip digit for digit
    stored = json.load(open(os.path.join(G, "ref_tail_demos.json")))["demos"]["FINAL_E_deformed"]["stored"]
    for k in ("volume_cm3", "contact_area_mm2", "max_depth_mm", "force_N", "mm_per_px", "estimated_grating_period_px", "depth_eps_mm"):
        assert back[k] == stored[k]
    cp = pkg.write_result_csv(str(tmp_path), rec)
    rows = list(csv.reader(open(cp)))
    assert rows[0] == ["reference_path", "deformed_path", "volume_cm3", "force_N", "contact_area_mm2", "max_depth_mm", "mm_per_px",
                       "estimated_grating_period_px", "ftp_output_dir", "force_model_type"]
    assert rows[1][-1] == fm["type"] and float(rows[1][2]) == res["volume_cm3"]
    masks = {k: rng.random((n, n)) > 0.5 for k in ("roi_eroded", "reliable", "output_reliable", "circ_mask", "contact_kept_by_depth",
                                                   "hole_candidates", "contact_dilated")}
    b = pkg.height_map_bundle(height, masks, (10, 20, 10 + n, 20 + n), (100, 120), (26, 36, 15), (16, 16, 15))
    keys = list(b.keys())
    assert keys[:2] == ["height_crop", "height_full"]
    assert keys[2:9] == ["crop_" + k for k in masks] and keys[9:16] == ["full_" + k for k in masks]
    assert keys[16:] == ["meta_crop_x1", "meta_crop_y1", "meta_crop_x2", "meta_crop_y2", "meta_roi_center_x_full", "meta_roi_center_y_full",
                         "meta_roi_radius_full", "meta_roi_center_x_crop", "meta_roi_center_y_crop", "meta_roi_radius_crop"]
    assert b["height_full"].shape == (100, 120) and b["height_full"].dtype == np.float32 and np.isnan(b["height_full"][0, 0])
    assert np.array_equal(b["height_full"][20:20 + n, 10:10 + n], height, equal_nan=True)
    assert b["full_reliable"].dtype == bool and np.array_equal(b["full_reliable"][20:20 + n, 10:10 + n], masks["reliable"])
    assert b["meta_crop_x1"].dtype == np.int32 and b["meta_crop_x1"].shape == ()
    paths = pkg.export_heightmap_files(str(tmp_path), b)
    z = np.load(paths["bundle_npz"])
    assert sorted(z.files) == sorted(keys) and np.array_equal(z["height_crop"], height, equal_nan=True)
    assert np.array_equal(np.load(paths["crop_npy"]), height, equal_nan=True)
    first = open(paths["crop_csv"]).readline().strip().split(",")
    assert len(first) == n and first[0] == "nan"


def test_force_calibration_fit_reproduces_the_stored_model(pkg):
    """N4 host logic: the six candidate curves fitted to the reference's own per-image table (75 rows, stored
    `per_image_results.csv`) give the stored `calibration_model.json`: same best model, parameters, RMSE ranking."""
    import csv
    rows = list(csv.DictReader(open(os.path.join(G, "ref_per_image_results.csv"))))
    assert len(rows) == 75 and [float(r["force_N"]) for r in rows[:6]] == [0.5] * 5 + [1.0]
    stored = json.load(open(os.path.join(G, "calibration_height_to_force.json")))
    model = pkg.calibrate.calibration_model(rows, stored["reference_path"], stored["deformed_dir"], stored["output_dir"])
    assert list(model.keys()) == list(stored.keys())
    for k in ("volume_definition", "grating_pitch_mm", "depth_eps_mm", "anchor_origin", "origin_weight"):
        assert model[k] == stored[k]
    b, sb = model["best_model"], stored["best_model"]
    assert list(b.keys()) == list(sb.keys())
    assert b["type"] == sb["type"] == "growth" and b["n_fit"] == 95 and b["n_samples"] == 75
    for k in ("a", "b"):
        assert abs(b["params"][k] - sb["params"][k]) <= 1e-5 * abs(sb["params"][k])          # scipy trust-region fit, same start values
    assert abs(b["rmse"] - sb["rmse"]) <= 1e-8 * sb["rmse"] and abs(b["r2"] - sb["r2"]) <= 1e-8
    assert b["equation"] == sb["equation"]
    assert [c["type"] for c in model["candidates_summary"]] == [c["type"] for c in stored["candidates_summary"]]
    for c, sc in zip(model["candidates_summary"], stored["candidates_summary"]):
        assert abs(c["rmse"] - sc["rmse"]) <= 1e-6 * sc["rmse"], c["type"]
    # prediction through the fitted model equals the C-ABI force curve
    v = 0.11378655442935222
    assert abs(float(pkg.calibrate.model_predict(b, v)) - pkg.predict_force_from_volume(b, v)) <= 1e-12


def test_phase_to_height_fit_reproduces_the_stored_model(pkg, tmp_path):
    """N4, second calibrator (Code/phase_to_height.py:1280-1383, :1491-1545): the four stored rows of the reference's
    `Force/Phase_to_height/calibration_out/calibration_results.csv` (tests/golden/ref_phase_to_height_results.csv) give the stored
    `calibration_model.json` (tests/golden/calibration_phase_to_height.json): same best model, parameters, RMSE ranking; the CSV
    writer reproduces the stored file byte for byte; and the rows the path oracle computes from the four calibration photographs
    (tests/golden/e2e_phase_to_height_report.json) give the same curve within the measurement differences."""
    import csv
    C = pkg.calibrate
    stored_csv = os.path.join(G, "ref_phase_to_height_results.csv")
    rows = list(csv.DictReader(open(stored_csv)))
    assert [r["file"] for r in rows] == [f for f, _ in C.PHASE_TO_HEIGHT_SAMPLES]
    assert [float(r["depth_mm"]) for r in rows] == [d for _, d in C.PHASE_TO_HEIGHT_SAMPLES]
    stored = json.load(open(os.path.join(G, "calibration_phase_to_height.json")))
    model = C.phase_to_height_model(rows, stored["reference_path"], stored["deformed_dir"], stored["output_dir"])
    assert list(model.keys()) == list(stored.keys())
    sb, mb = stored["best_model"], model["best_model"]
    assert list(mb.keys()) == list(sb.keys()) and mb["type"] == sb["type"] == "hinge_saturating" and mb["n"] == sb["n"] == 4
    for k in ("a", "b"):
        assert abs(mb["params"][k] - sb["params"][k]) <= 1e-5 * abs(sb["params"][k]), (k, mb["params"][k], sb["params"][k])
    assert abs(mb["params"]["c"] - sb["params"]["c"]) <= 1e-6          # the hinge sits at the origin (stored: -1.8e-9)
    assert abs(mb["rmse"] - sb["rmse"]) <= 1e-6 and abs(mb["r2"] - sb["r2"]) <= 1e-5
    assert [c["type"] for c in model["candidates_summary"]] == [c["type"] for c in stored["candidates_summary"]]
    for cm, cs in zip(model["candidates_summary"], stored["candidates_summary"]):
        assert abs(cm["rmse"] - cs["rmse"]) <= 1e-5 * max(1.0, cs["rmse"]), (cm, cs)
    for k in ("use_negated_height_for_fit", "x_definition", "interpretation", "reference_path", "deformed_dir", "output_dir"):
        assert model[k] == stored[k]
    # the curve the path consumes: same millimetres over the range of the samples
    x = np.linspace(0.0, 1.3, 50)
    assert np.max(np.abs(C.model_predict(mb, x) - C.model_predict(sb, x))) <= 1e-5
    # CSV writer: byte-identical to the stored file when fed the stored values
    typed = [{"file": r["file"], "depth_mm": float(r["depth_mm"]), "min_height_unitless": float(r["min_height_unitless"]),
              "min_x": int(r["min_x"]), "min_y": int(r["min_y"]), "heightmap_figure": r["heightmap_figure"]} for r in rows]
    out = C.write_phase_to_height_csv(str(tmp_path / "calibration_results.csv"), typed)
    assert open(out, "rb").read() == open(stored_csv, "rb").read()
    # end to end: minima the path oracle measured on the four photographs -> the same curve within their differences (<= 6e-4 unitless)
    rep = json.load(open(os.path.join(G, "e2e_phase_to_height_report.json")))
    rows2 = [{"file": r["file"], "depth_mm": d, "min_height_unitless": r["min"]} for r, (_, d) in zip(rep, C.PHASE_TO_HEIGHT_SAMPLES)]
    m2 = C.phase_to_height_model(rows2, "", "", "")["best_model"]
    assert m2["type"] == "hinge_saturating"
    assert np.max(np.abs(C.model_predict(m2, x) - C.model_predict(sb, x))) <= 5e-3          # mm, against 1.9-2.1 mm samples
    # non-finite rows are skipped, fewer than two valid samples is an error (phase_to_height.py:1485-1492)
    with pytest.raises(RuntimeError):
        C.phase_to_height_model([{"file": "a", "depth_mm": 1.0, "min_height_unitless": float("nan")},
                                 {"file": "b", "depth_mm": 2.0, "min_height_unitless": -1.0}], "", "", "")
    assert pkg.FtpConfig.phase_to_height().roi_erode_px == 80 and pkg.FtpConfig.phase_to_height().plane_order_for_removal == 0


def test_multimodal_summary_reproduces_the_stored_file(pkg, tmp_path):
    """N1 remainder: combined_outputs/multimodal_summary.json (Code/multimodal_sensor.py:592-650).  From the readings of the stored FINAL_E
    session and the four calibration JSONs of the reference tree (fixtures: data files), the writer gives the stored file back -- same keys,
    same order, same values, byte for byte."""
    stored_txt = open(os.path.join(G, "ref_multimodal_summary_FINAL_E.json"), encoding="utf-8").read()
    stored = json.loads(stored_txt)
    load = lambda f: json.load(open(os.path.join(G, f), encoding="utf-8"))
    f, t = stored["sensor_readings"]["force"], stored["sensor_readings"]["temperature"]
    res = {"force_N": f["force_N"], "volume_cm3": f["volume_cm3"], "contact_area_mm2": f["contact_area_mm2"], "max_depth_mm": f["max_depth_mm"],
           "mm_per_px": f["scale_mm_per_px"]}
    s = pkg.multimodal_summary(stored["session_id"], stored["timestamp"], stored["input_images"]["reference"], stored["input_images"]["deformed"],
                               stored["output_directory"], res, t, load("calibration_phase_to_height.json"), load("calibration_height_to_force.json"),
                               load("ref_temp_color_metrics.json"), load("ref_temp_black_metrics.json"), stored["file_paths"]["force_subdir"],
                               stored["file_paths"]["temperature_subdir"], stored["file_paths"]["combined_subdir"])
    assert s == stored
    path = pkg.write_multimodal_summary(str(tmp_path), s)
    assert open(path, encoding="utf-8").read() == stored_txt
    # a missing calibration file gives an empty block, as upstream's load_json_safe -> None does
    s2 = pkg.multimodal_summary("x", "y", "r", "d", "o", res, t, None, None, None, None, "a", "b", "c")
    assert s2["calibration_performance"] == {"phase_to_height": {}, "height_to_force": {}, "temperature_color_model": {}, "temperature_black_model": {}}
    # temperature statistics block (multimodal_sensor.py:558-567)
    tm = np.array([[20.0, 21.0], [np.nan, 25.0]], np.float32)
    st = pkg.temperature_statistics(tm, np.isfinite(tm))
    assert list(st) == ["mean_C", "median_C", "std_C", "min_C", "max_C", "valid_pixels"] and st["valid_pixels"] == 3 and st["median_C"] == 21.0
    assert np.isnan(pkg.temperature_statistics(tm, np.zeros((2, 2), bool))["mean_C"])


def test_polyfit_kernels_restore_no_undefined_sgpr_spill_lane(tmp_path):
    """The column polyfit kernels are the only ones in the library that spill hundreds of SGPRs into VGPR lanes (and, for the register-capped
    variant, those VGPRs on to scratch).  A GPU memory fault of an experimental 512-thread variant in round 2 was never reproduced; what can be
    checked without running anything is that the compiler's spill code is consistent: tools/check_sgpr_spill_lanes.py runs a must-be-defined
    analysis over the kernel's control-flow graph for every (spill VGPR, lane) and (scratch slot, lane) pair.  Both shipped RP = 56 kernels
    must come out with 0 findings (so does the re-instantiated 512-thread variant)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, PKG_NAME, "csrc")
    asm = str(tmp_path / "k_fit.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-I", csrc,
                           os.path.join(csrc, "k_fit.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    for name in ("_ZN2vf20k_robust_polyfit_colILi56ELi4EEEvPKfPKhiifiiPfS5_iiii", "_ZN2vf23k_robust_polyfit_col_w5ILi56ELi4EEEvPKfPKhiifiiPfS5_iiii"):
        out = subprocess.check_output([__import__("sys").executable, os.path.join(ROOT, "tools", "check_sgpr_spill_lanes.py"), asm, name], text=True)
        assert out.strip().endswith("0 findings"), out[-2000:]
