"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (BASELINE.json north_star): height maps within 1e-4 relative to the map's peak value
(`|gpu - oracle| <= 1e-4 * max|oracle|`, float32 path), scalars within 1e-4 relative, the
arg-extremum ("contact-location") index bit-exact, byte masks equal.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ftp_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-4


_FORCE_MODEL = [None]


@pytest.fixture(scope="module")
def cal(pkg):
    model, neg = pkg.load_calibration(os.path.join(G, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(G, "calibration_height_to_force.json"))["best_model"]
    _FORCE_MODEL[0] = fm
    return model, neg, fm


def _sensor(pkg, cal, n, cfg, max_batch, ref=None, **kw):
    ref = pkg.synth.reference_frame(n, **kw) if ref is None else ref
    return ref, pkg.FtpSensor(ref, pkg.synth.roi_circle(n), cfg, cal[0], cal[1], cal[2], max_batch=max_batch)


def _check_frame(out, b, o, n, check_argmin=True):
    """compare GPU outputs of frame b against oracle result dict o.  RTOL (1e-4) is relative to the map's PEAK (a depth map is zero over
    most of the ROI, a per-pixel relative bar is meaningless there): every pixel within 1e-4 * max|map|; scalars within 1e-4 relative."""
    hm = out["height_map_mm"][b].cpu().numpy()
    ref = o["height_map_mm_crop"]
    assert int(out["status"][b]) == 0
    assert np.array_equal(np.isnan(hm), np.isnan(ref))
    peak = max(float(np.nanmax(np.abs(ref))), 1e-6)
    diff = np.abs(hm - ref)
    assert float(np.nanmax(diff)) <= RTOL * peak
    rel = out["output_reliable"][b].cpu().numpy().astype(bool)
    mism = int((rel != o["output_reliable_crop"]).sum())
    assert mism == 0
    s = out["scalars"][b].cpu().numpy()
    assert int(s[4]) == o["argmax_depth_index"]                       # arg-max depth index: bit-exact
    v, (ax, ay) = o["argmin_unitless"]
    if check_argmin:
        assert int(s[8]) == ay * n + ax                              # arg-min unitless index: bit-exact
    assert abs(s[7] - v) <= RTOL * max(abs(v), 1e-6)
    for i, key in ((0, "volume_cm3"), (1, "contact_area_mm2"), (2, "max_depth_mm")):
        assert abs(s[i] - o[key]) <= RTOL * max(abs(o[key]), 1e-9), key
    # The force is a FUNCTION of the volume (force_sensor.py:149-167; shipped: growth a(e^{bV} - 1)): a volume inside the bar maps to a force
    # inside the bar times the curve's condition number |V f'(V) / f(V)| (1.8 at V = 0.14 cm^3), so that is the force's bar.
    V, F = o["volume_cm3"], o["force_N"]
    if V > 0 and abs(F) > 0 and _FORCE_MODEL[0] is not None:
        dV = 1e-6 * V
        kappa = abs((O.predict_force_from_volume(_FORCE_MODEL[0], V + dV) - O.predict_force_from_volume(_FORCE_MODEL[0], V - dV)) / (2 * dV) * V / F)
    else:
        kappa = 1.0
    # the widening is capped (a genuine force-tail error must not hide behind a large condition number), and the curve
    # evaluation itself is checked tightly on the GPU's OWN volume, so that only the volume carries the 1e-4 bar.
    assert abs(s[3] - F) <= RTOL * min(4.0, max(1.0, kappa)) * max(abs(F), 1e-9), "force_N"
    if _FORCE_MODEL[0] is not None:
        assert abs(s[3] - O.predict_force_from_volume(_FORCE_MODEL[0], float(s[0]))) <= 1e-12 * max(1.0, abs(s[3])), "force curve on the GPU's own volume"
    # the carrier peak is refined in float32 from spectrum magnitudes: equal to ~1 ulp of the peak position
    assert abs(s[5] - o["estimated_grating_period_px"]) <= 1e-5 * s[5] and abs(s[6] - o["mm_per_px"]) <= 1e-5 * s[6]
    assert int(s[9]) == int(o["reliable"].sum())


@pytest.mark.parametrize("mode", ["scaled", "shipped"])
def test_full_path_matches_oracle_224(pkg, cal, mode):
    n, nb = 224, 6
    cfg = pkg.FtpConfig.scaled(n) if mode == "scaled" else pkg.FtpConfig.as_shipped()
    ref, sensor = _sensor(pkg, cal, n, cfg, nb)
    frames = pkg.synth.deformed_batch(n, 0, nb)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    info = sensor.reference_info
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    # carrier search: same integer bin; the float64 log-parabolic refinement agrees to the rounding of log() (device vs glibc)
    assert np.allclose(info["peak_refined"], rs["demod"]["peak_refined"], rtol=0, atol=1e-9)
    assert info["fft_shape"] == rs["demod"]["fft_shape"]
    for b in range(nb):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        _check_frame(out, b, o, n)
        s = out["scalars"][b].cpu().numpy()
        it = o["inter"]
        # exact order statistics + NumPy float32 interpolation: thresholds agree to float32 resolution
        assert abs(s[11] - it["amp_thr"]) <= 2e-6 * abs(it["amp_thr"])
        assert abs(s[13] - it["bg_med"]) <= 2e-6          # median of an O(1 rad) plane: a few float32 ulps
        assert int(s[14]) == int(it["demod"]["inter"]["bad"].sum())
        assert bool(s[10]) == o["flipped"]


def test_stage_intermediates_224(pkg, cal):
    """Stage by stage.  Planes that feed a hard threshold (Sobel magnitude / intensity -> bad-pixel mask, amplitude product -> Gaussian
    -> quality >= p25) must equal the oracle's BIT FOR BIT: the demodulation runs in float64 up to the float32 amplitude / phase (as the
    oracle's complex128 transform) and the Gaussian executes cv's row / symmetric-column sequence with fused multiply-adds (as
    oracle/cvlite.c).  Continuous planes downstream of the float32 LAPACK lstsq keep a tolerance."""
    n, nb = 224, 2
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb)
    sensor._test_set("keep_planes", 1)
    frames = pkg.synth.deformed_batch(n, 40, nb)
    sensor.predict_batch(frames)
    torch.cuda.synchronize()
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    P = n * n

    def plane(name, dt=torch.float32):
        return sensor.intermediate(name, nb, dt).cpu().numpy()

    cref = sensor.intermediate("cref", 1, torch.float64).cpu().numpy().reshape(n, n, 2)
    assert np.abs((cref[..., 0] + 1j * cref[..., 1]) - rs["demod"]["field"]).max() <= 1e-12 * np.abs(rs["demod"]["field"]).max()
    roi = rs["roi"]

    def same_bits_in_roi(a, b):
        """bit-identical inside the ROI; outside it the apodised field is ~1e-7 of its scale inside, so the ~1e-15 (relative to that
        scale) difference between the two float64 transforms shows as last-bit differences of values nothing ever reads"""
        return np.array_equal(a[roi], b[roi]) and float(np.abs(a - b).max()) <= 1e-9 * float(np.abs(b).max())

    assert same_bits_in_roi(sensor.intermediate("amp_ref", 1).cpu().numpy().reshape(n, n), rs["demod"]["amp"])
    for b in range(nb):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        it, sl = o["inter"], slice(b * P, (b + 1) * P)
        di = it["demod"]["inter"]
        assert np.array_equal(plane("bad1", torch.uint8)[sl].reshape(n, n) != 0, di["bad"])          # Sobel, percentiles, dilate
        assert plane("thr_hi")[b] == np.float32(di["hi_thr"]) and plane("thr_g")[b] == np.float32(di["g_thr"])
        inp = plane("img")[sl].reshape(n, n)
        assert np.abs(inp - di["img_inpainted"]).max() <= 1e-5 * 255                                   # Telea inpaint
        fld = plane("field", torch.float64).reshape(nb, n, n, 2)[b]
        fo = it["demod"]["field"]
        assert np.abs((fld[..., 0] + 1j * fld[..., 1]) - fo).max() <= 1e-12 * np.abs(fo).max()         # pruned DFT == fft2/ifft2 path (complex128)
        assert np.array_equal(inp, di["img_inpainted"])                                                # ... in OpenCV's summation order: same bits
        assert same_bits_in_roi(plane("amp")[sl].reshape(n, n), it["demod"]["amp"])                    # |field| -> float32
        assert same_bits_in_roi(plane("quality")[sl].reshape(n, n), it["quality"])                     # blur(amp_ref * amp_def)
        wg, wo = plane("wrapped")[sl].reshape(n, n), it["wrapped"]                                     # angle(cdef conj(cref)) -> float32: an
        assert int((wg != wo)[o["reliable"]].sum()) <= 1e-3 * P and np.abs(wg - wo)[o["reliable"]].max() <= 1e-7   # absolute angle error 1e-15 / |field| vs ulp(angle)
        assert float(sensor._last_out["scalars"][b, 11]) == np.float32(it["amp_thr"])                 # p25 threshold
        assert np.array_equal(plane("rel0", torch.uint8)[sl].reshape(n, n) != 0, it["thresholded"])
        assert np.array_equal(plane("reliable", torch.uint8)[sl].reshape(n, n) != 0, o["reliable"])   # close, CC, chamfer erode
        uw, uo = plane("unwrapped")[sl].reshape(n, n), it["unwrapped"]
        assert np.array_equal(np.isnan(uw), np.isnan(uo))
        m = ~np.isnan(uo)
        assert np.abs(uw[m] - uo[m]).max() <= 1e-5
        assert np.array_equal(plane("contact_d", torch.uint8)[sl].reshape(n, n) != 0, o["contact_dilated"])
        assert np.array_equal(plane("background", torch.uint8)[sl].reshape(n, n) != 0, it["background"])
        assert np.array_equal(plane("kept", torch.uint8)[sl].reshape(n, n) != 0, o["contact_kept_by_depth"])
        hu, ho = plane("unitless")[sl].reshape(n, n), o["height_unitless"]
        assert np.array_equal(np.isnan(hu), np.isnan(ho))
        assert np.nanmax(np.abs(hu - ho)) <= RTOL * np.nanmax(np.abs(ho))


def test_unwrap_with_true_wraps_and_parent_tree(pkg, cal):
    """Deformations of several 2*pi: the flood must reproduce the oracle's wrap counts exactly."""
    n, nb = 160, 3
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=5)
    frames = pkg.synth.deformed_batch(n, 0, nb, config=5, amp_scale=9.0)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    P = n * n
    uw_all = sensor.intermediate("unwrapped", nb).cpu().numpy().copy()
    wr_all = sensor.intermediate("wrapped", nb).cpu().numpy().copy()
    # these frames are path-independent on the reliable mask: the consistency check (k_unwrap_fast.hip) settled them without the flood ...
    assert (sensor.intermediate("unwrap_need", nb, torch.int32).cpu().numpy() == 0).all()
    # ... and the priority flood (check off) gives the same plane bit for bit, plus the spanning tree compared below
    sensor._test_set("unwrap_fast", 0)
    out_flood = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert np.array_equal(sensor.intermediate("unwrapped", nb).cpu().numpy(), uw_all, equal_nan=True)
    assert np.array_equal(out_flood["height_map_mm"].cpu().numpy(), out["height_map_mm"].cpu().numpy(), equal_nan=True)
    par_all = sensor.intermediate("parent", nb, torch.int32).cpu().numpy()
    saw_wrap = False
    for b in range(nb):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        it = o["inter"]
        uo = it["unwrapped"]
        uw = uw_all[b * P:(b + 1) * P].reshape(n, n)
        wr = wr_all[b * P:(b + 1) * P].reshape(n, n)
        assert np.array_equal(np.isnan(uw), np.isnan(uo))
        m = ~np.isnan(uo)
        k_gpu = np.rint((uw[m] - wr[m]) / (2 * np.pi))
        k_ora = np.rint((uo[m] - it["wrapped"][m]) / (2 * np.pi))
        assert np.array_equal(k_gpu, k_ora)
        saw_wrap |= bool(np.any(k_ora != 0))
        assert np.abs(uw[m] - uo[m]).max() <= 2e-5
        # the spanning tree itself (parent of every pixel) equals the reference's heap order
        from oracle import cvlite
        _, par_o, _ = cvlite.unwrap_quality_guided(it["wrapped"], o["reliable"], it["quality"], want_tree=True)
        qg = sensor.intermediate("quality", nb).cpu().numpy()[b * P:(b + 1) * P].reshape(n, n)
        if np.array_equal(qg, it["quality"]):       # identical keys -> identical tree, bit for bit
            assert np.array_equal(par_all[b * P:(b + 1) * P].reshape(n, n), par_o)
        _check_frame(out, b, o, n)
    assert saw_wrap


def test_big_frame_flood_reproduces_the_heap_order(pkg, cal):
    """Frames beyond the uint16 rank range (here 320 x 320 = 103 684 padded pixels; the native crops have 1.4 M) take the bitmap flood of
    k_unwrap_big.hip: 32-bit ranks, priority queue as a three-level bitmap in LDS, plane in global memory, up to 8 pops per step.
    Deformations of several 2*pi; the parent of every pixel must equal the reference heap's, bit for bit, and the generic one-pop
    kernel (test hook flood_tier = 0) must give the same tree."""
    n, nb = 320, 2
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=5)
    frames = pkg.synth.deformed_batch(n, 0, nb, config=5, amp_scale=9.0)
    # with the consistency check on (the default) these path-independent frames never reach the flood; its plane is kept for comparison
    sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert (sensor.intermediate("unwrap_need", nb, torch.int32).cpu().numpy() == 0).all()
    uw_check = sensor.intermediate("unwrapped", nb).cpu().numpy().copy()
    sensor._test_set("unwrap_fast", 0)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert np.array_equal(sensor.intermediate("unwrapped", nb).cpu().numpy(), uw_check, equal_nan=True)
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    P = n * n
    uw_all = sensor.intermediate("unwrapped", nb).cpu().numpy().copy()
    wr_all = sensor.intermediate("wrapped", nb).cpu().numpy().copy()
    par_all = sensor.intermediate("parent", nb, torch.int32).cpu().numpy().copy()
    qg_all = sensor.intermediate("quality", nb).cpu().numpy().copy()
    from oracle import cvlite
    saw_wrap = False
    for b in range(nb):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        it = o["inter"]
        uo = it["unwrapped"]
        uw = uw_all[b * P:(b + 1) * P].reshape(n, n)
        wr = wr_all[b * P:(b + 1) * P].reshape(n, n)
        assert np.array_equal(np.isnan(uw), np.isnan(uo))
        m = ~np.isnan(uo)
        k_gpu = np.rint((uw[m] - wr[m]) / (2 * np.pi))
        k_ora = np.rint((uo[m] - it["wrapped"][m]) / (2 * np.pi))
        assert np.array_equal(k_gpu, k_ora)
        saw_wrap |= bool(np.any(k_ora != 0))
        _, par_o, _ = cvlite.unwrap_quality_guided(it["wrapped"], o["reliable"], it["quality"], want_tree=True)
        assert np.array_equal(qg_all[b * P:(b + 1) * P].reshape(n, n)[rs["roi"]], it["quality"][rs["roi"]])
        assert np.array_equal(par_all[b * P:(b + 1) * P].reshape(n, n), par_o), b
        _check_frame(out, b, o, n)
    assert saw_wrap
    for tier in (0, 3):       # 0: generic kernel only; 3: bitmap flood hands every frame back (the path of masks larger than its bitmap)
        sensor._test_set("flood_tier", tier)
        sensor.predict_batch(frames)
        torch.cuda.synchronize()
        assert np.array_equal(sensor.intermediate("parent", nb, torch.int32).cpu().numpy(), par_all), tier
        assert np.array_equal(sensor.intermediate("unwrapped", nb).cpu().numpy(), uw_all, equal_nan=True), tier


def test_input_formats_agree(pkg, cal):
    n, nb = 128, 2
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb)
    g = pkg.synth.deformed_batch(n, 7, nb)
    base = sensor.predict_batch(g)["height_map_mm"].clone()
    bgr = np.repeat(g[..., None], 3, axis=-1)
    for frames in (torch.from_numpy(g).to(torch.float16), bgr, torch.from_numpy(bgr).to(torch.float16)):
        got = sensor.predict_batch(frames)["height_map_mm"]
        assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(base, nan=-1.0))
    # a genuinely coloured frame goes through the fixed-point BGR2GRAY of OpenCV 4.x
    rng = np.random.default_rng(3)
    col = np.clip(bgr.astype(np.int32) + rng.integers(-6, 7, size=bgr.shape), 0, 255).astype(np.uint8)
    gray = ((col[..., 0].astype(np.int64) * 3735 + col[..., 1].astype(np.int64) * 19235 + col[..., 2].astype(np.int64) * 9798 + (1 << 14)) >> 15).astype(np.uint8)
    a = sensor.predict_batch(col)["height_map_mm"].clone()
    b = sensor.predict_batch(gray)["height_map_mm"]
    assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))


def test_edge_cases(pkg, cal):
    n = 128
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, 4)
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    # deformed == reference: zero phase everywhere -> no contact
    out = sensor.predict_batch(ref[None])
    o = O.process_frame(ref, rs, cfg, *cal)
    # a flat (noise-level) height field has no meaningful arg-min: index not compared here
    _check_frame(out, 0, o, n, check_argmin=False)
    assert float(out["scalars"][0, 3]) == o["force_N"]
    # featureless frame: upstream returns None when the reliable mask is empty (shape_ftp.py:1677-1679)
    blank = np.full((n, n), 90, np.uint8)
    # (a constant frame demodulates to pure rounding noise, so its map is not comparable; the call must
    #  complete with a valid status and the NaN-outside-ROI layout)
    ob = sensor.predict_batch(blank[None])
    torch.cuda.synchronize()
    assert int(ob["status"][0]) in (0, 1)
    hm_blank = ob["height_map_mm"][0].cpu().numpy()
    assert np.all(np.isnan(hm_blank[~rs["roi"]]))
    assert int(ob["status"][0]) == 1 or not np.any(np.isnan(hm_blank[rs["roi"]]))
    # error behaviour of the boundary
    with pytest.raises(RuntimeError):
        sensor.predict_batch(np.zeros((1, n + 2, n), np.uint8))          # size mismatch (shape_ftp.py:1477)
    with pytest.raises(RuntimeError):
        sensor.predict_batch(np.zeros((5, n, n), np.uint8))              # batch > max_batch
    with pytest.raises(ValueError):
        pkg.FtpSensor(ref, pkg.synth.roi_circle(n), cfg, {"type": "nope", "params": {}}, True, cal[2], max_batch=1)
    with pytest.raises(ValueError):
        sensor.predict_batch(np.zeros((1, n, n), np.float32))
    # single-frame API returns the reference's result dict
    r1 = sensor.predict(pkg.synth.deformed_frame(n, 1))
    assert set(["height_map_mm_crop", "roi_eroded_crop", "output_reliable_crop", "estimated_grating_period_px"]) <= set(r1)
    assert r1["height_map_mm_crop"].dtype == np.float32 and r1["roi_eroded_crop"].dtype == bool
    assert np.array_equal(r1["roi_eroded_crop"], rs["roi"])


def test_force_tail_on_device(pkg):
    g = np.load(os.path.join(G, "ref_numpy_small.npz"))
    h = g["tail_h"]
    for sign, key in ((1, "tail_res"), (-1, "tail_res_neg")):
        got = pkg.depth_map_to_volume_cm3(sign * h, np.isfinite(h), 0.0303784, 0.01)
        assert got[1] == g[key][1] and got[2] == g[key][2]               # area, max depth: exact
        assert abs(got[0] - g[key][0]) <= 1e-6 * g[key][0]               # float32 sum order
    assert pkg.depth_map_to_volume_cm3(h * 0, np.isfinite(h), 0.0303784, 0.01) == (0.0, 0.0, 0.0)
    d = np.load(os.path.join(G, "ref_tail_demo_E_small.npz"))
    got = pkg.depth_map_to_volume_cm3(d["height"], None, float(d["mm_per_px"][0]), 0.01)   # roi = isfinite (multimodal_sensor.py:388)
    assert got[1] == d["tail"][1] and got[2] == d["tail"][2]
    assert abs(got[0] - d["tail"][0]) <= 1e-6 * d["tail"][0]
    batch = np.stack([h, -h, h * 0])
    res = pkg.depth_map_to_volume_cm3(batch, None, 0.0303784, 0.01)
    assert res.shape == (3, 3) and res[2].tolist() == [0.0, 0.0, 0.0]


def test_full_batch_properties_256(pkg, cal):
    """BASELINE size (batch 256 at 224x224): determinism, batch-position independence, replicated frames."""
    n, B = 224, 256
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, B)
    base = pkg.synth.deformed_batch(n, 100, 16)
    frames = torch.from_numpy(np.concatenate([base] * 16, axis=0)).cuda()
    o1 = {k: v.clone() for k, v in sensor.predict_batch(frames).items()}
    o2 = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert int((o1["status"] != 0).sum()) == 0
    for k in ("height_map_mm", "output_reliable", "scalars"):
        a, b = o1[k], o2[k]
        if a.is_floating_point():
            a, b = torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)
        assert torch.equal(a, b), k                                     # run-to-run bit identical
    hm = torch.nan_to_num(o1["height_map_mm"], nan=-7.0)
    for rep in range(1, 16):
        assert torch.equal(hm[rep * 16:(rep + 1) * 16], hm[:16])         # position in the batch does not matter
    small = sensor.predict_batch(frames[3:5])
    assert torch.equal(torch.nan_to_num(small["height_map_mm"], nan=-7.0), hm[3:5])
    # spot-check three frames of the big batch against the oracle
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    for b in (0, 7, 15):
        _check_frame(o1, 240 + b, O.process_frame(base[b], rs, cfg, *cal), n)


def test_native_size_1182_as_shipped(pkg, cal):
    """Native crop size of the reference (1182x1182, constants exactly as shipped): exercises the
    global-memory fallbacks of the sequential kernels (frames too large for the LDS-resident paths)."""
    n = 1182
    cfg = pkg.FtpConfig.as_shipped()
    ref, sensor = _sensor(pkg, cal, n, cfg, 1, config=7)
    frame = pkg.synth.deformed_frame(n, 0, config=7)
    out = sensor.predict_batch(frame[None])
    torch.cuda.synchronize()
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    pg, po = sensor.reference_info["peak_refined"], rs["demod"]["peak_refined"]
    assert (round(pg[0]), round(pg[1])) == (round(po[0]), round(po[1]))          # same integer carrier bin
    assert abs(pg[0] - po[0]) <= 1e-9 and abs(pg[1] - po[1]) <= 1e-9             # float64 log-parabolic refinement
    o = O.process_frame(frame, rs, cfg, *cal)
    _check_frame(out, 0, o, n)
    # batches of 16 native crops and more take the one-wave two-pass chamfer (k_chamfer2<20>) for the 200 px frontier band instead of
    # the closed form: forced here on the single frame (test hook `chamfer_twopass`), same map and scalars
    hm0, sc0 = out["height_map_mm"].cpu().numpy().copy(), out["scalars"].cpu().numpy().copy()
    sensor._test_set("chamfer_twopass", 1)
    alt = sensor.predict_batch(frame[None])
    torch.cuda.synchronize()
    assert int(alt["status"][0]) == 0
    assert np.array_equal(alt["height_map_mm"].cpu().numpy(), hm0, equal_nan=True)
    assert np.array_equal(alt["scalars"].cpu().numpy(), sc0)


def test_non_square_odd_sizes(pkg, cal):
    """h != w, odd dimensions, ROI circle off-centre and touching no border."""
    h, w = 151, 203
    cfg = pkg.FtpConfig.scaled(160)
    period = 65.83619546657023 * 160 / 1182
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)

    def mk(phi, seed):
        r = np.random.default_rng(seed)
        img = 128.0 * (1.0 + 0.1 * np.cos(xx / 60.0)) * (0.55 + 0.35 * np.cos(2 * np.pi * xx / period + phi)) + r.normal(0, 2.0, (h, w))
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    ref = mk(0.0, 1)
    frames = np.stack([mk(-0.8 * np.exp(-((xx - 90 - 10 * i) ** 2 + (yy - 70) ** 2) / (2 * 18.0 ** 2)), 2 + i) for i in range(2)])
    circle = (98, 74, 66)
    sensor = pkg.FtpSensor(ref, circle, cfg, cal[0], cal[1], cal[2], max_batch=2)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    rs = O.make_reference_state(ref, *circle, cfg)
    assert np.allclose(sensor.reference_info["peak_refined"], rs["demod"]["peak_refined"], rtol=0, atol=1e-9)
    for b in range(2):
        o = O.process_frame(frames[b], rs, cfg, *cal)
        hm = out["height_map_mm"][b].cpu().numpy()
        assert hm.shape == (h, w)
        ref_hm = o["height_map_mm_crop"]
        assert np.array_equal(np.isnan(hm), np.isnan(ref_hm))
        assert np.nanmax(np.abs(hm - ref_hm)) <= RTOL * max(np.nanmax(np.abs(ref_hm)), 1e-6)
        s = out["scalars"][b].cpu().numpy()
        assert int(s[4]) == o["argmax_depth_index"]
        assert int(out["status"][b]) == 0
    # the cluster front end on the same odd-sized frames (LDS cluster windows + the padded global planes of k_inpaint_big.hip, both queue
    # variants): same inpainted plane, same map
    img0 = sensor.intermediate("img", 2).cpu().numpy().copy()
    hm0 = out["height_map_mm"].cpu().numpy().copy()
    sensor._test_set("inpaint_tier", 0)
    for lds in (1, 0):
        sensor._test_set("big_queue_lds", lds)
        alt = sensor.predict_batch(frames)
        torch.cuda.synchronize()
        assert (alt["status"].cpu().numpy() == 0).all()
        assert np.array_equal(sensor.intermediate("img", 2).cpu().numpy(), img0)
        assert np.array_equal(alt["height_map_mm"].cpu().numpy(), hm0, equal_nan=True)



@pytest.mark.parametrize("var,val", [("inpaint_tier", 1), ("inpaint_tier", 0), ("flood_tier", 1), ("flood_tier", 0), ("chamfer_twopass", 1), ("unwrap_fast", 0)])
def test_alternative_kernel_tiers_agree(pkg, cal, var, val):
    """The fallback / opt-in kernels (whole-frame and cluster-parallel Telea, one-pop and scan floods, two-pass chamfer) stay
    parity-green: same frames through the default path and through the alternative, compared with each other and with the oracle."""
    n, nb = 224, 6
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=3)
    frames = pkg.synth.deformed_batch(n, 40, nb, config=3)
    if var == "flood_tier":
        sensor._test_set("unwrap_fast", 0)                                  # the flood tiers are compared tree by tree: every frame through the flood
    base = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    hm0 = base["height_map_mm"].cpu().numpy().copy()
    par0 = sensor.intermediate("parent", nb, torch.int32).cpu().numpy().copy()
    img0 = sensor.intermediate("img", nb).cpu().numpy().copy()
    sensor._test_set(var, val)                                              # csrc/test_hooks.h: per-session, not an environment switch
    alt = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    hm1 = alt["height_map_mm"].cpu().numpy().copy()
    par1 = sensor.intermediate("parent", nb, torch.int32).cpu().numpy().copy()
    st = alt["status"].cpu().numpy()
    assert (st == 0).all()
    if var == "unwrap_fast":
        assert np.array_equal(hm0, hm1, equal_nan=True)                     # consistency check + parallel integration against the priority flood
    elif var != "inpaint_tier":
        if var == "flood_tier":
            assert np.array_equal(par0, par1)                               # the growth tree is an integer result: identical
        assert np.array_equal(hm0, hm1, equal_nan=True)
    else:
        # every Telea tier (frame window, whole frame, cluster by cluster in LDS windows / on the global planes) pops in the queue's order and
        # sums the estimator in OpenCV's order: same inpainted plane, bit for bit, hence the same map
        assert np.array_equal(sensor.intermediate("img", nb).cpu().numpy(), img0)
        assert np.array_equal(hm0, hm1, equal_nan=True)
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    _check_frame(alt, 0, O.process_frame(frames[0], rs, cfg, *cal), n)


def test_big_cluster_march_lds_and_global_queue_agree_with_the_window_march(pkg, cal):
    """k_inpaint_big.hip (clusters no LDS window takes: padded global planes, per-pop code of the LDS windows) with its queue in LDS and
    -- test hook `big_queue_lds` = 0 -- in the wave's slice of global memory: the inpainted plane of the cluster front end
    (`inpaint_tier` = 0; at 224 x 224 the hole pixels are one cluster, far beyond the 3072-cell cluster windows) equals the frame-window
    march's bit for bit in both variants."""
    n, nb = 224, 12
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=3)
    frames = pkg.synth.deformed_batch(n, 300, nb, config=3)
    sensor.predict_batch(frames)
    torch.cuda.synchronize()
    img0 = sensor.intermediate("img", nb).cpu().numpy().copy()
    sensor._test_set("inpaint_tier", 0)
    for lds in (1, 0):
        sensor._test_set("big_queue_lds", lds)
        out = sensor.predict_batch(frames)
        torch.cuda.synchronize()
        assert (out["status"].cpu().numpy() == 0).all()
        assert np.array_equal(sensor.intermediate("img", nb).cpu().numpy(), img0), "queue in %s" % ("LDS" if lds else "global memory")


@pytest.mark.parametrize("n,nb,shipped", [(224, 48, False), (1182, 2, True)])
def test_sessions_in_flight_give_the_same_bits_as_alone(pkg, cal, n, nb, shipped):
    """bench.py's default keeps three sessions in flight, each on its own stream with its own workspace.  Nothing of a session may depend on
    what else runs on the chip: three sessions with different batches, issued round-robin on three streams for several rounds, must return
    exactly what each returns alone -- at bench size (window march, one workgroup per frame everywhere) and on native crops (cluster
    march with 128 KB of LDS per wave, chains of kernels over the batch)."""
    cfg = pkg.FtpConfig() if shipped else pkg.FtpConfig.scaled(n)
    sensors, frames, alone = [], [], []
    for k in range(3):
        _, s = _sensor(pkg, cal, n, cfg, nb, config=3)
        f = torch.from_numpy(pkg.synth.deformed_batch(n, 700 + 100 * k, nb, config=3)).cuda()
        o = s.predict_batch(f)
        torch.cuda.synchronize()
        assert (o["status"].cpu().numpy() == 0).all()
        sensors.append(s); frames.append(f)
        alone.append({key: o[key].cpu().numpy().copy() for key in ("height_map_mm", "scalars", "output_reliable")})
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [None] * 3
    for rnd in range(4):
        for k in range(3):
            with torch.cuda.stream(streams[k]):
                outs[k] = sensors[k].predict_batch(frames[k], outs[k])
    torch.cuda.synchronize()
    for k in range(3):
        for key in alone[k]:
            assert np.array_equal(outs[k][key].cpu().numpy(), alone[k][key], equal_nan=True), (k, key)


def test_phase_to_height_constants_variant(pkg, cal):
    """The constants of the reference's offline calibrator (Code/phase_to_height.py:63, :115, no debug_ramp): ROI erosion, a wider
    frontier band and no plane pre-removal, scaled to 224: same parity bar as the default configuration."""
    n = 224
    cfg = pkg.FtpConfig.scaled(n)
    cfg.roi_erode_px = int(round(80 * n / 1182))
    cfg.frontier_zero_band_px = int(round(300 * n / 1182))
    cfg.plane_order_for_removal = 0
    ref, sensor = _sensor(pkg, cal, n, cfg, 2, config=3)
    frames = pkg.synth.deformed_batch(n, 7, 2, config=3)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    for b in range(2):
        _check_frame(out, b, O.process_frame(frames[b], rs, cfg, *cal), n)


def test_inpaint_window_tier_against_sequential_tier_and_oracle_on_many_frames(pkg, cal):
    """The window march executes up to four outside-pass pops per step and a straight-line fill (k_inpaint_win.hip); both must
    leave the pop order of the one-at-a-time march untouched.  48 frames with different hole layouts through the window tier and
    through the whole-frame sequential tier (test hook `inpaint_tier` = 1, one pop per step): the inpainted planes must agree to the
    float-sum tolerance (the tiers reduce the estimator sums in different orders; a pop out of order moves a filled pixel by far
    more), and a sample of the frames is checked against the oracle's OpenCV-restated Telea."""
    n, nb = 224, 48
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=3)
    frames = np.concatenate([pkg.synth.deformed_batch(n, 200, nb // 2, config=3), pkg.synth.deformed_batch(n, 500, nb // 2, config=3, amp_scale=0.5)])
    P = n * n
    sensor.predict_batch(frames)
    torch.cuda.synchronize()
    img0 = sensor.intermediate("img", nb).cpu().numpy().reshape(nb, n, n).copy()
    bad0 = sensor.intermediate("bad1", nb, torch.uint8).cpu().numpy().reshape(nb, n, n).copy()
    sensor._test_set("inpaint_tier", 1)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    img1 = sensor.intermediate("img", nb).cpu().numpy().reshape(nb, n, n).copy()
    assert (out["status"].cpu().numpy() == 0).all()
    assert bad0.any(axis=(1, 2)).all()                                   # every frame has holes to fill
    assert np.array_equal(img0[bad0 == 0], img1[bad0 == 0])              # known pixels untouched, bit for bit
    assert float(np.abs(img0 - img1).max()) <= 1e-5 * 255
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    for b in (0, 11, 23, 24, 37, 47):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        di = o["inter"]["demod"]["inter"]
        assert np.array_equal(bad0[b] != 0, di["bad"])
        assert np.array_equal(img0[b], di["img_inpainted"]), b             # the window tiers sum in OpenCV's order: same bits
    # the 16-wave first tier (ordering pass + dataflow fills, k_inpaint_mw.hip) against the single-wave window march: same bits on every frame
    sensor._test_set("inpaint_tier", 2)
    sensor._test_set("telea_mw", 0)
    sensor.predict_batch(frames)
    torch.cuda.synchronize()
    img2 = sensor.intermediate("img", nb).cpu().numpy().reshape(nb, n, n)
    assert np.array_equal(img0, img2)


def test_march_lds_tiers_retry_and_whole_frame_fallback(pkg, cal):
    """The window march runs in three tiers (k_inpaint_win.hip): 110.75 KB of LDS for windows of up to 10 752 cells, the full-size
    retry (14 464 cells) for the frames the first tier hands back, the whole-frame kernel for the rest.  Saturated spots far apart
    stretch the bounding window of the hole pixels: frame 1 lands in the retry tier, frame 2 in the whole-frame kernel, frame 0
    stays in the first tier.  All three must equal the oracle at the strict bar, the inpainted planes bit for bit with the two-tier
    front end on and off."""
    n = 224
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, 3, config=3)
    frames = pkg.synth.deformed_batch(n, 300, 3, config=3).copy()
    yy, xx = np.mgrid[0:n, 0:n]
    for b, spots in ((1, ((66, 66), (160, 158))), (2, ((45, 47), (182, 178)))):
        for (cy, cx) in spots:
            frames[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= 9] = 255
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    P = n * n
    bad = sensor.intermediate("bad1", 3, torch.uint8).cpu().numpy().reshape(3, n, n) != 0
    img_a = sensor.intermediate("img", 3).cpu().numpy().reshape(3, n, n).copy()
    rng_px = int(round(cfg.bad_inpaint_radius))
    cells = []
    for b in range(3):
        ys, xs = np.where(bad[b])
        cells.append((ys.max() - ys.min() + 1 + 2 * (rng_px + 1)) * (xs.max() - xs.min() + 1 + 2 * (rng_px + 1)))
    assert cells[0] <= 10752 < cells[1] <= 14464 < cells[2], cells
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    for b in range(3):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        di = o["inter"]["demod"]["inter"]
        assert np.array_equal(bad[b], di["bad"]), b
        assert np.array_equal(img_a[b], di["img_inpainted"]), b            # OpenCV's summation order in every tier: same bits
        _check_frame(out, b, o, n)
    sensor._test_set("telea_two_tier", 0)
    sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert np.array_equal(sensor.intermediate("img", 3).cpu().numpy().reshape(3, n, n), img_a)
    for two_tier in (1, 0):                      # and without the 16-wave first tier (frame 0 then marches on the single-wave tiers)
        sensor._test_set("telea_mw", 0)
        sensor._test_set("telea_two_tier", two_tier)
        sensor.predict_batch(frames)
        torch.cuda.synchronize()
        assert np.array_equal(sensor.intermediate("img", 3).cpu().numpy().reshape(3, n, n), img_a)


def test_sixty_four_more_frames_against_oracle(pkg, cal):
    """Breadth: 64 further synthetic frames (two amplitude scales) end to end against the oracle, EVERY frame at the strict bar of the
    other tests (map within 1e-4 of its peak, masks equal, arg-max index equal).  Guards the data-dependent kernels (run-based
    components, chamfer-ball erosion, exact selection, batched march / flood) against inputs the small tests do not reach."""
    n, nb = 224, 64
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=3)
    frames = np.concatenate([pkg.synth.deformed_batch(n, 1000, nb // 2, config=3), pkg.synth.deformed_batch(n, 2000, nb // 2, config=3, amp_scale=1.6)])
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    rel = sensor.intermediate("reliable", nb, torch.uint8).cpu().numpy().reshape(nb, n, n) != 0
    qual = sensor.intermediate("quality", nb).cpu().numpy().reshape(nb, n, n)
    for b in range(nb):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        assert np.array_equal(qual[b][rs["roi"]], o["inter"]["quality"][rs["roi"]]), b   # the plane the reliable mask is thresholded from: same bits
        assert np.array_equal(rel[b], o["reliable"]), b
        _check_frame(out, b, o, n)


def _moat_frame(pkg, n, seed, ref=False, ir=18, mw=12, neck=7, ox=40):
    """A fringe frame whose contrast drops to 8 % on a ring ("moat") around an island that stays connected to the rest through a narrow
    corridor: the reliable mask is ONE component after largest_connected_component, and erode_by_distance then cuts the corridor, so the
    quality-guided flood never reaches the island (shape_ftp.py:770-773, :1043-1080)."""
    rng = np.random.default_rng(seed)
    p = pkg.synth.NATIVE_PERIOD_PX * n / pkg.synth.NATIVE_CROP
    cx, cy, r = pkg.synth.roi_circle(n)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    s = 1.0 + 0.15 * np.cos(np.pi * np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2) / r)
    contrast, phi = np.ones((n, n)), 0.0
    if not ref:
        d = np.sqrt((xx - (cx + ox)) ** 2 + (yy - cy) ** 2)
        contrast[(d >= ir) & (d <= ir + mw) & ~((np.abs(yy - cy) <= neck / 2.0) & (xx < cx + ox))] = 0.08
        phi = -0.7 * np.exp(-((xx - cx + 20) ** 2 + (yy - cy - 10) ** 2) / (2 * (0.12 * n) ** 2))
    img = 128.0 * s * (0.55 + 0.35 * contrast * np.cos(2 * np.pi * xx / p + phi)) + rng.normal(0, 2.0, (n, n))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_internal_holes_branch_without_reliable_smoothing(pkg, cal):
    """§8(a) row 15: compute_internal_holes_within_mask + inpaint_only_mask (shape_ftp.py:1153-1204, :1770-1801).  The branch is only live
    with RELIABLE_SMOOTH_SIGMA_PX = 0 (a positive sigma leaves every reliable pixel finite).  Frames whose reliable mask falls apart in
    erode_by_distance leave an island the flood never reaches: NaN inside reliable -> hole candidates (box-filter fraction, chamfer
    distance) -> Telea fill (radius 5) of the candidates, the rest leaves output_reliable.  Hole constants chosen so that all three
    outcomes (filled, dropped, untouched) occur; a wider demodulation patch (20 bins) resolves the 12-pixel moat."""
    n = 224
    cfg = pkg.FtpConfig.scaled(n)
    cfg.reliable_smooth_sigma_px = 0.0
    cfg.reliable_edge_margin_px = 7
    cfg.valid_close_kernel = 3
    cfg.patch_half_width_bins = 20
    cfg.hole_neighborhood_px, cfg.hole_known_fraction, cfg.hole_min_dist_px = 71, 0.25, 1
    ref = _moat_frame(pkg, n, 99999, ref=True)
    frames = np.stack([_moat_frame(pkg, n, 7), _moat_frame(pkg, n, 8, ir=16, mw=12, neck=9), pkg.synth.deformed_frame(n, 3)])
    nb = len(frames)
    _, sensor = _sensor(pkg, cal, n, cfg, nb, ref=ref)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    holes_g = sensor.intermediate("hole_cand", nb, torch.uint8).cpu().numpy().reshape(nb, n, n) != 0
    rel_g = sensor.intermediate("reliable", nb, torch.uint8).cpu().numpy().reshape(nb, n, n) != 0
    seen_filled = seen_dropped = 0
    for b in range(nb):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        assert np.array_equal(rel_g[b], o["reliable"])
        assert np.array_equal(holes_g[b], o["hole_candidates"])
        _check_frame(out, b, o, n)                                   # includes output_reliable == oracle's, map within 1e-4 of the peak
        seen_filled += int(o["hole_candidates"].sum())
        seen_dropped += int((o["reliable"] & ~o["output_reliable_crop"]).sum())
        if b == 0:
            assert np.array_equal(sensor.masks(0)["hole_candidates"], o["hole_candidates"])
    assert seen_filled > 100 and seen_dropped > 10                   # the branch really ran: pixels filled AND pixels dropped
    # with the smoothing on, the same frames never enter the branch: output_reliable == reliable upstream and here
    cfg2 = pkg.FtpConfig.scaled(n)
    cfg2.reliable_edge_margin_px, cfg2.valid_close_kernel, cfg2.patch_half_width_bins = 7, 3, 20
    _, s2 = _sensor(pkg, cal, n, cfg2, nb, ref=ref)
    out2 = s2.predict_batch(frames)
    torch.cuda.synchronize()
    rs2 = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg2)
    for b in range(nb):
        o2 = O.process_frame(frames[b], rs2, cfg2, *cal)
        assert not o2["hole_candidates"].any() and np.array_equal(o2["output_reliable_crop"], o2["reliable"])
        _check_frame(out2, b, o2, n)


def test_uncached_pairs_128(pkg, cal):
    """BASELINE configs[4] as SURVEY.md 8(d) restates it: (reference, deformed) PAIRS with the reference-frame demodulation NOT cached --
    carrier search + reference demodulation per sample, as Code/height_to_force.py:384 runs shape_ftp.main per image.  128 pairs = the
    per-GPU shard of 1024 / 8.  Every sample has its OWN reference frame (different noise realisation, two grating periods), checked
    against O.make_reference_state + O.process_frame per pair at the strict bar."""
    n, nb = 224, 128
    cfg = pkg.FtpConfig.scaled(n)
    periods = [pkg.synth.NATIVE_PERIOD_PX * n / pkg.synth.NATIVE_CROP, 11.3]
    refs = np.stack([pkg.synth._base(n, 0.0, np.random.default_rng(770000 + b), periods[b % 2]) for b in range(nb)])
    defs = np.stack([pkg.synth.deformed_frame(n, 4000 + b, config=3, period=periods[b % 2], amp_scale=1.0 + 0.5 * (b % 3)) for b in range(nb)])
    sensor = pkg.FtpSensor(None, pkg.synth.roi_circle(n), cfg, cal[0], cal[1], cal[2], max_batch=nb, frame_shape=(n, n))
    out = sensor.predict_pairs(refs, defs)
    torch.cuda.synchronize()
    info = sensor.pair_info(nb)
    assert int((out["status"] != 0).sum()) == 0
    for b in range(nb):
        rs = O.make_reference_state(refs[b], *pkg.synth.roi_circle(n), cfg)
        assert np.allclose(info[b]["peak_refined"], rs["demod"]["peak_refined"], rtol=0, atol=1e-9), b
        _check_frame(out, b, O.process_frame(defs[b], rs, cfg, *cal), n)
    # the same pairs through a session per reference (cached mode) give the same bits: one code path builds the tables of both modes
    for b in (0, 1, 77):
        _, s1 = _sensor(pkg, cal, n, cfg, 1, ref=refs[b])
        o1 = s1.predict_batch(defs[b][None])
        assert torch.equal(torch.nan_to_num(o1["height_map_mm"][0], nan=-7.0), torch.nan_to_num(out["height_map_mm"][b], nan=-7.0))
        assert torch.equal(o1["scalars"][0], out["scalars"][b])
    # with room for both frame sets in the workspace (max_batch >= 2 B) the two preprocessing passes run as one: same bits
    s2 = pkg.FtpSensor(None, pkg.synth.roi_circle(n), cfg, cal[0], cal[1], cal[2], max_batch=2 * nb, frame_shape=(n, n))
    ob = s2.predict_pairs(refs, defs)
    torch.cuda.synchronize()
    assert torch.equal(ob["status"], out["status"]) and torch.equal(ob["scalars"], out["scalars"])
    assert torch.equal(torch.nan_to_num(ob["height_map_mm"], nan=-7.0), torch.nan_to_num(out["height_map_mm"], nan=-7.0))
    assert torch.equal(ob["output_reliable"], out["output_reliable"])
    # a featureless reference frame (its "carrier" is rounding noise) must not disturb its neighbours in the batch
    refs2, defs2 = refs[:4].copy(), defs[:4].copy()
    refs2[2] = 90
    o2 = sensor.predict_pairs(refs2, defs2)
    torch.cuda.synchronize()
    st = o2["status"].cpu().numpy()
    assert (st[[0, 1, 3]] == 0).all() and st[2] in (0, 1, 3)
    for b in (0, 1, 3):
        assert torch.equal(torch.nan_to_num(o2["height_map_mm"][b], nan=-7.0), torch.nan_to_num(out["height_map_mm"][b], nan=-7.0))


def test_module_level_predict_is_the_drop_in(pkg, cal):
    """`predict(image, reference=...) -> force map` is the API name BASELINE.json asks for (it does not exist upstream: it is shape_ftp.main's
    result dict, Code/shape_ftp.py:2029-2037, plus the force tail of Code/multimodal_sensor.py:388-419).  First call builds the session from the
    reference frame; later calls reuse it; a new reference rebuilds it."""
    n = 160
    cfg = pkg.FtpConfig.scaled(n)
    circle = pkg.synth.roi_circle(n)
    ref = pkg.synth.reference_frame(n, config=3)
    kw = dict(roi_circle=circle, config=cfg, height_model=cal[0], use_negated_height=cal[1], force_model=cal[2])
    rs = O.make_reference_state(ref, *circle, cfg)
    for i, image in enumerate([pkg.synth.deformed_frame(n, 11), pkg.synth.deformed_frame(n, 12)]):
        res = pkg.predict(image, reference=ref, **kw) if i == 0 else pkg.predict(image)         # second call: cached session
        o = O.process_frame(image, rs, cfg, *cal)
        assert set(["height_map_mm_crop", "roi_eroded_crop", "output_reliable_crop", "estimated_grating_period_px", "force_N", "volume_cm3",
                    "argmax_depth_index"]) <= set(res)
        hm, r = res["height_map_mm_crop"], o["height_map_mm_crop"]
        assert hm.dtype == np.float32 and np.array_equal(np.isnan(hm), np.isnan(r))
        assert float(np.nanmax(np.abs(hm - r))) <= RTOL * float(np.nanmax(np.abs(r)))
        assert np.array_equal(res["output_reliable_crop"], o["output_reliable_crop"]) and np.array_equal(res["roi_eroded_crop"], o["roi_eroded_crop"])
        assert res["argmax_depth_index"] == o["argmax_depth_index"]
        assert abs(res["force_N"] - o["force_N"]) <= RTOL * max(abs(o["force_N"]), 1e-9)
        assert abs(res["estimated_grating_period_px"] - o["estimated_grating_period_px"]) <= 1e-9 * o["estimated_grating_period_px"]
    # a new reference frame rebuilds the session
    ref2 = pkg.synth.reference_frame(n, config=3, period=10.1)
    img2 = pkg.synth.deformed_frame(n, 13, period=10.1)
    res2 = pkg.predict(img2, reference=ref2, **kw)
    o2 = O.process_frame(img2, O.make_reference_state(ref2, *circle, cfg), cfg, *cal)
    assert res2["argmax_depth_index"] == o2["argmax_depth_index"]
    assert float(np.nanmax(np.abs(res2["height_map_mm_crop"] - o2["height_map_mm_crop"]))) <= RTOL * float(np.nanmax(np.abs(o2["height_map_mm_crop"])))
    with pytest.raises(RuntimeError):
        pkg.predict(np.zeros((n + 1, n), np.uint8))                                            # size mismatch (shape_ftp.py:1477)


def _vortex_frame(pkg, n, seed, x0, y0, charge=1):
    """A deformed frame whose fringes fork at (x0, y0): phi gains charge * atan2(y - y0, x - x0), a phase residue inside the ROI.  The wrapped
    phase difference then sums to 2 pi * charge around the fork, so the unwrapped plane depends on the spanning tree the flood grows."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    phi = -0.6 * np.exp(-((xx - 0.4 * n) ** 2 + (yy - 0.55 * n) ** 2) / (2.0 * (0.15 * n) ** 2)) + charge * np.arctan2(yy - y0, xx - x0)
    return pkg.synth._base(n, phi, rng)


def test_unwrap_consistency_check_falls_back_on_a_residue(pkg, cal):
    """k_unwrap_fast.hip settles a frame only when every 8-adjacent pair of the seed's component satisfies k[b] - k[a] = c(a -> b); then
    any spanning tree gives the same plane.  Frames 0 and 2 carry a fringe fork (a phase residue) inside the reliable mask: the check must
    refuse them (need = 1) and the priority flood must produce the reference's tree-dependent plane -- same wrap counts and same parents
    as the oracle's heap order; frame 1 is an ordinary frame and is settled by the check (need = 0).  The whole batch is then repeated with
    the check off: identical planes."""
    n, nb = 224, 3
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=3)
    frames = np.stack([_vortex_frame(pkg, n, 77001, 0.62 * n, 0.40 * n), pkg.synth.deformed_frame(n, 3, config=3), _vortex_frame(pkg, n, 77002, 0.35 * n, 0.60 * n, -1)])
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    need = sensor.intermediate("unwrap_need", nb, torch.int32).cpu().numpy()
    assert need.tolist() == [1, 0, 1]
    P = n * n
    uw_all = sensor.intermediate("unwrapped", nb).cpu().numpy().copy()
    wr_all = sensor.intermediate("wrapped", nb).cpu().numpy().copy()
    par_all = sensor.intermediate("parent", nb, torch.int32).cpu().numpy().copy()
    qg_all = sensor.intermediate("quality", nb).cpu().numpy().copy()
    hm = out["height_map_mm"].cpu().numpy().copy()
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    from oracle import cvlite
    for b in range(nb):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        it = o["inter"]
        uo = it["unwrapped"]
        uw = uw_all[b * P:(b + 1) * P].reshape(n, n)
        wr = wr_all[b * P:(b + 1) * P].reshape(n, n)
        assert np.array_equal(np.isnan(uw), np.isnan(uo)), b
        m = ~np.isnan(uo)
        k_gpu = np.rint((uw[m] - wr[m]) / (2 * np.pi))
        k_ora = np.rint((uo[m] - it["wrapped"][m]) / (2 * np.pi))
        assert np.array_equal(k_gpu, k_ora), b
        if need[b]:
            assert np.any(k_ora != 0), b                       # the residue forces a 2 pi cut somewhere in the mask
            assert np.array_equal(qg_all[b * P:(b + 1) * P].reshape(n, n)[rs["roi"]], it["quality"][rs["roi"]])
            _, par_o, _ = cvlite.unwrap_quality_guided(it["wrapped"], o["reliable"], it["quality"], want_tree=True)
            assert np.array_equal(par_all[b * P:(b + 1) * P].reshape(n, n), par_o), b
    sensor._test_set("unwrap_fast", 0)
    out2 = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert np.array_equal(sensor.intermediate("unwrapped", nb).cpu().numpy(), uw_all, equal_nan=True)
    assert np.array_equal(out2["height_map_mm"].cpu().numpy(), hm, equal_nan=True)


def test_unwrap_consistency_check_against_flood_on_many_frames(pkg, cal):
    """96 frames (three amplitude scales, genuine 2 pi wraps among them) through the consistency check + parallel integration and through the
    priority flood: the unwrapped planes must be the same bits, frame by frame, NaN layout included."""
    n, nb = 224, 96
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=3)
    frames = np.concatenate([pkg.synth.deformed_batch(n, 4000, 32, config=3), pkg.synth.deformed_batch(n, 4100, 32, config=3, amp_scale=9.0),
                             pkg.synth.deformed_batch(n, 4200, 32, config=3, amp_scale=0.4)])
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    need = sensor.intermediate("unwrap_need", nb, torch.int32).cpu().numpy()
    uw = sensor.intermediate("unwrapped", nb).cpu().numpy().copy()
    wr = sensor.intermediate("wrapped", nb).cpu().numpy().copy()
    hm = out["height_map_mm"].cpu().numpy().copy()
    assert (need == 0).sum() >= nb // 2, need.tolist()         # the check settles these frames (all of them, in practice)
    fin = np.isfinite(uw)
    assert np.any(np.rint((uw[fin] - wr[fin]) / (2 * np.pi)) != 0)   # the batch does contain wrapped pixels
    sensor._test_set("unwrap_fast", 0)
    out2 = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert np.array_equal(sensor.intermediate("unwrapped", nb).cpu().numpy(), uw, equal_nan=True)
    assert np.array_equal(out2["height_map_mm"].cpu().numpy(), hm, equal_nan=True)
    assert (out["status"].cpu().numpy() == out2["status"].cpu().numpy()).all()


def test_march_generation_overflow_is_handed_back(pkg, cal):
    """The 16-wave march holds at most 2 048 entries per FMM generation (k_inpaint_mw.hip); a lattice of saturated dots gives ~120 separate 5 x 5
    blobs in one window, i.e. more band pixels than that in generation 0.  Such a frame must come back from the single-wave tiers with the
    oracle's plane, bit for bit, like any other frame -- with the 16-wave tier on and off."""
    n = 224
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, 2, config=3)
    frames = pkg.synth.deformed_batch(n, 700, 2, config=3).copy()
    for y in range(70, 158, 8):
        for x in range(68, 156, 8):
            frames[1][y, x] = 255
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    img = sensor.intermediate("img", 2).cpu().numpy().reshape(2, n, n).copy()
    bad = sensor.intermediate("bad1", 2, torch.uint8).cpu().numpy().reshape(2, n, n) != 0
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    for b in range(2):
        o = O.process_frame(frames[b], rs, cfg, *cal, keep_intermediates=True)
        di = o["inter"]["demod"]["inter"]
        assert np.array_equal(bad[b], di["bad"]), b
        assert np.array_equal(img[b], di["img_inpainted"]), b
        _check_frame(out, b, o, n)
    from oracle import cvlite
    nseed = int((cvlite.dilate(bad[1].astype(np.uint8), np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)) > 0).sum() - bad[1].sum())
    assert nseed > 2048, nseed                                  # the frame does exceed a generation of the 16-wave tier
    sensor._test_set("telea_mw", 0)
    sensor.predict_batch(frames)
    torch.cuda.synchronize()
    assert np.array_equal(sensor.intermediate("img", 2).cpu().numpy().reshape(2, n, n), img)


def test_large_frame_chains_agree_with_one_workgroup_per_frame(pkg, cal):
    """Frames of 512 x 512 and more take k_big.hip: exact selections and IRLS fits as chains of streaming kernels over all pixels of the batch
    instead of one workgroup per frame.  The selections are exact in both forms -- thresholds and medians must be the SAME bits --, the fits
    sum their normal equations in a different (fixed) order, so the maps may differ by float64 summation noise (1e-6 of the peak at most);
    both forms against the oracle at the usual bar."""
    n, nb = 512, 2
    cfg = pkg.FtpConfig.scaled(n)
    ref, sensor = _sensor(pkg, cal, n, cfg, nb, config=3)
    frames = pkg.synth.deformed_batch(n, 40, nb, config=3)
    out = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    hm0 = out["height_map_mm"].cpu().numpy().copy()
    sel0 = {k: sensor.intermediate(k, nb).cpu().numpy().copy() for k in ("thr_hi", "thr_g", "mu", "thr3", "core_thr", "core_med")}
    rs = O.make_reference_state(ref, *pkg.synth.roi_circle(n), cfg)
    o0 = O.process_frame(frames[0], rs, cfg, *cal)
    _check_frame(out, 0, o0, n)
    sensor._test_set("big_chain", 0)
    out1 = sensor.predict_batch(frames)
    torch.cuda.synchronize()
    hm1 = out1["height_map_mm"].cpu().numpy()
    _check_frame(out1, 0, o0, n)
    assert np.array_equal(np.isnan(hm0), np.isnan(hm1))
    assert float(np.nanmax(np.abs(hm0 - hm1))) <= 1e-6 * float(np.nanmax(np.abs(hm1)))
    for k in ("thr_hi", "thr_g", "mu"):                      # selections upstream of the fits: identical inputs, exact order statistics
        assert np.array_equal(sel0[k], sensor.intermediate(k, nb).cpu().numpy()), k
    for k in ("thr3", "core_thr", "core_med"):               # downstream of the fits: same to the fits' summation noise
        a, b = sel0[k], sensor.intermediate(k, nb).cpu().numpy()
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7), k
