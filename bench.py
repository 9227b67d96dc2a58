#!/usr/bin/env python
"""Benchmark of the MI355X-native VISTAF image -> force-map path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (vistaf_ftp_predict_batch: gray -> bad-pixel inpaint -> illumination
normalise -> pruned-DFT demodulation -> reliable mask -> quality-guided unwrap -> robust detrend -> smoothing ->
frontier/compose -> mm curve -> blob filter -> force tail) over one batch of synthetic 224x224x3 fp16 frames that
are already resident in HBM, followed (N > 1) by the single RCCL all-gather of the outputs.
Workload = BASELINE.json configs[2]: batch 256 per GPU (weak scaling: N GPUs process N*256 frames per step).
Steps are issued round-robin to `--inflight` sessions (default 3), each with its own HIP stream and workspace, the way a
serving loop would keep the GPU busy: the march kernels of the path run one wave per frame and leave most of a CU idle,
which the other session's kernels fill.  `--inflight 1` gives strictly serial steps; `stage_ms` / `roofline` are always
measured on one session alone.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "vistaf-roboskin-vision-integrated-multimodal-sensor_amd"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _cpu_worker(args):
    """one process of the CPU baseline: frames[i0::stride] through the oracle until the budget is spent -> (frames done, seconds)"""
    n, constants, i0, stride, budget_s, pairs = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    pkg = importlib.import_module(PKG)
    from oracle import ftp_oracle as O
    cfg = pkg.FtpConfig.scaled(n) if constants == "scaled" else pkg.FtpConfig.as_shipped()
    g = os.path.join(ROOT, "tests", "golden")
    cal, neg = pkg.load_calibration(os.path.join(g, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(g, "calibration_height_to_force.json"))["best_model"]
    circle = pkg.synth.roi_circle(n)
    ref = pkg.synth.reference_frame(n, config=3)
    rs = None if pairs else O.make_reference_state(ref, *circle, cfg)
    done, i, busy = 0, i0, 0.0
    while busy < budget_s:
        f = pkg.synth.deformed_frame(n, i, config=3)   # frame synthesis is not timed
        t0 = time.perf_counter()
        if pairs:                                     # uncached pair: the reference frame is demodulated again for every sample
            rs = O.make_reference_state(ref, *circle, cfg)
        O.process_frame(f, rs, cfg, cal, neg, fm)
        busy += time.perf_counter() - t0
        done += 1
        i += stride
    return done, busy


def _usable_cores():
    """cores this process may really use: the scheduler affinity capped by the cgroup CPU quota (a GPU box hands each job a share of the host)"""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, min(cores, 64))          # (64 worker processes are plenty for a baseline figure; the GPU box allows few processes per job)


def cpu_baselines(n, constants, budget_s, pairs=False):
    """The oracle (NumPy + C restatement of the reference's path; kind "port") timed on this host BEFORE the GPU is initialised (the
    worker processes are forked): (a) single process, frames sequential, as the reference's only batch driver does
    (Code/height_to_force.py:360), figure rendering excluded; (b) one process per host core, so that the GPU / CPU ratio is
    not flattered by single-threading."""
    import multiprocessing as mp
    what = "uncached (reference, deformed) pairs" if pairs else "frames, reference-frame demodulation cached and excluded"
    done, dt = _cpu_worker((n, constants, 0, 1, budget_s, pairs))
    one = {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} synthetic {n}x{n} {what}; sequential, single process on 1 of {_usable_cores()} usable host cores ({os.cpu_count()} on the host), {dt:.1f} s; "
                     f"figure rendering excluded"}
    cores = _usable_cores()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(n, constants, k, cores, budget_s * 0.75, pairs) for k in range(cores)])
    tot = sum(r[0] / r[1] for r in res)
    allc = {"value": tot, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{sum(r[0] for r in res)} synthetic {n}x{n} {what}; {cores} processes (one per host core, BLAS/OpenMP threads = 1), "
                      f"{max(r[1] for r in res):.1f} s each; sum of the per-process rates"}
    return one, allc


GATHER_KEYS = ("height_map_mm", "scalars", "status")


class Stepper:
    """One benchmark step = one pass of the path over one batch on the next session in turn (its own stream and workspace), followed, when
    `gathers` is given (N > 1), by the step's ONE collective on the same stream.  Kept free of GPU specifics so that the world-size-2 gloo
    test (tests/test_host_cpu.py) drives exactly this control flow on CPU tensors."""

    def __init__(self, run, sensors, outs, streams, gathers, stream_ctx):
        self.run, self.sensors, self.outs, self.streams, self.gathers, self.stream_ctx = run, sensors, outs, streams, gathers, stream_ctx
        self.n = 0

    def step(self):
        k = self.n % len(self.sensors)
        self.n += 1
        if self.streams[k] is None:
            self.run(self.sensors[k], self.outs[k])
            if self.gathers is not None:
                self.gathers[k].gather(self.outs[k])
        else:
            with self.stream_ctx(self.streams[k]):
                self.run(self.sensors[k], self.outs[k])
                if self.gathers is not None:
                    self.gathers[k].gather(self.outs[k])


def timed_steps(stepper, warmup, steps, dist, sync, dev):
    """W untimed steps, then exactly K steps bracketed by barrier + device synchronisation on both sides; MAX over ranks."""
    for _ in range(warmup):
        stepper.step()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        stepper.step()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed


MFMA_F64_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: FP64 matrix (= FP64 vector) peak


def _dft_gemm_record(stage_ms, B, n, sensor, demods):
    cfg = sensor.config
    pw = 2 * max(3, int(cfg.patch_half_width_bins)) + 1
    fwd = 2.0 * (B * n) * n * (2 * pw)                 # [B*h x w] . [w x 2pw]
    inv = 2.0 * 2.0 * (B * n * n) * (2 * pw)           # [h x 2ph] . [2ph x {re, im} w] per frame
    flops = demods * (fwd + inv)
    tf = flops / (stage_ms * 1e-3) / 1e12 if stage_ms > 0 else 0.0
    return {"kernels": ["k_dft_fwd1_mfma", "k_dft_inv2_mfma"], "dtype": "f64", "flops_per_step": flops, "stage_ms": round(stage_ms, 4),
            "achieved_TFLOPs_over_stage": round(tf, 3), "peak_TFLOPs": MFMA_F64_PEAK_TFLOPS, "frac_of_mfma_peak_over_stage": round(tf / MFMA_F64_PEAK_TFLOPS, 4)}


def _csrc_sha():
    """fingerprint of the kernel sources: recorded by tools/collect_profiles.py next to the PMC traffic numbers, so that a traffic figure
    measured on other kernels is never reported as this run's"""
    import hashlib
    hsh = hashlib.sha1()
    d = os.path.join(ROOT, PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".h")):
            hsh.update(f.encode())
            hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def _traffic_table():
    """newest profiles/traffic_r*.json -> (table or {}, source description)"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    if not fs:
        return {}, None
    try:
        tj = json.load(open(fs[-1]))
    except Exception:
        return {}, None
    src = {"file": os.path.relpath(fs[-1], ROOT), "measured_at_head": tj.get("_head"), "date": tj.get("_date"), "csrc_sha": tj.get("_csrc_sha"),
           "matches_this_build": tj.get("_csrc_sha") == _csrc_sha(),
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_pmc.sh), not measured in this run"}
    return (tj if src["matches_this_build"] else {}), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--constants", choices=["scaled", "shipped"], default="scaled")
    ap.add_argument("--inflight", type=int, default=3,
                    help="sessions (each with its own HIP stream and workspace) that take the steps in turn, so consecutive steps overlap on the GPU; "
                         "1 = strictly serial steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the two extra figures measured after the timed region (constants as shipped, configs[1]): profiling runs")
    ap.add_argument("--kernel-tier", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostic A/B runs: kernel tier hooks of csrc/test_hooks.h (e.g. telea_two_tier=0); the default run sets none")
    ap.add_argument("--cpu-budget-s", type=float, default=16.0)
    ap.add_argument("--pairs", action="store_true",
                    help="BASELINE configs[4] as SURVEY.md 8(d) restates it: uncached (reference, deformed) pairs, carrier search + reference "
                         "demodulation per sample (no fusion of any kind exists upstream); default batch 128 = 1024 / 8 per GPU")
    args = ap.parse_args()
    if args.pairs and args.batch == 256:
        args.batch = 128

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_one = cpu_all = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_one, cpu_all = cpu_baselines(args.size, args.constants, args.cpu_budget_s, args.pairs)      # before any HIP call: forks workers
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # VISTAF_BENCH_REHEARSE=1: rehearse the N > 1 control flow on a one-GPU box (every rank on cuda:0, gloo instead of RCCL)
        rehearse = os.environ.get("VISTAF_BENCH_REHEARSE") == "1"
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    pkg = importlib.import_module(PKG)
    n, B = args.size, args.batch
    cfg = pkg.FtpConfig.scaled(n) if args.constants == "scaled" else pkg.FtpConfig.as_shipped()
    g = os.path.join(ROOT, "tests", "golden")
    cal, neg = pkg.load_calibration(os.path.join(g, "calibration_phase_to_height.json"))
    fm = pkg.load_force_calibration(os.path.join(g, "calibration_height_to_force.json"))["best_model"]

    # synthetic data: B DISTINCT deformed frames per rank (the sequential kernels last as long as their slowest frame,
    # so a tiled batch would under-sample that tail), 3-channel fp16
    ref = pkg.synth.reference_frame(n, config=3)
    frames_u8 = pkg.synth.deformed_batch(n, rank * B, B, config=3)
    frames = torch.from_numpy(frames_u8).to(dev)
    frames = frames[..., None].expand(-1, -1, -1, 3).to(torch.float16).contiguous()   # [B, n, n, 3] fp16
    ref3 = torch.from_numpy(ref).to(dev)[..., None].expand(-1, -1, 3).to(torch.float16).contiguous()
    if args.pairs:
        # every sample carries its own reference frame (own noise realisation): [B, n, n, 3] fp16 next to the deformed frames
        refs_u8 = np.stack([pkg.synth._base(n, 0.0, np.random.default_rng(880000 + rank * B + b)) for b in range(B)])
        refs = torch.from_numpy(refs_u8).to(dev)[..., None].expand(-1, -1, -1, 3).to(torch.float16).contiguous()
        # max_batch = 2 B: room for the reference and the deformed frames of a batch side by side, so both sets are preprocessed together
        sensors = [pkg.FtpSensor(None, pkg.synth.roi_circle(n), cfg, cal, neg, fm, max_batch=2 * B, device=dev, frame_shape=(n, n))
                   for _ in range(max(1, args.inflight))]
        run = lambda s_, o_=None: s_.predict_pairs(refs, frames, o_)
    else:
        sensors = [pkg.FtpSensor(ref3, pkg.synth.roi_circle(n), cfg, cal, neg, fm, max_batch=B, device=dev) for _ in range(max(1, args.inflight))]
        run = lambda s_, o_=None: s_.predict_batch(frames, o_)
    for kv in args.kernel_tier:
        name, _, val = kv.partition("=")
        for s_ in sensors:
            s_._test_set(name, int(val))
    streams = [torch.cuda.Stream(device=dev) for _ in sensors] if len(sensors) > 1 else [None]
    sensor = sensors[0]
    outs = [run(s_) for s_ in sensors]
    out = outs[0]
    torch.cuda.synchronize(dev)
    # N > 1: ONE RCCL all-gather per step -- maps, scalar records and status packed into one record per frame (parallel.PackedGather),
    # issued on the session's own stream so that it overlaps the other sessions' kernels
    gathers = [pkg.parallel.PackedGather(o, keys=GATHER_KEYS) for o in outs] if world > 1 else None
    stepper = Stepper(run, sensors, outs, streams, gathers, torch.cuda.stream)
    elapsed = timed_steps(stepper, args.warmup, args.steps, dist if world > 1 else None, lambda: torch.cuda.synchronize(dev), dev)
    status_bad = int((out["status"] != 0).sum().item())

    # per-kernel device times (HIP events recorded by the library on the launch stream), separate passes
    sensor.enable_stage_timing(True)
    acc = {}
    reps = 5
    for _ in range(reps):
        run(sensor, out)
        for k, v in sensor.stage_times_ms().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    sensor.enable_stage_timing(False)

    # SURVEY.md 8(d) asks for two more figures next to the headline; both are measured AFTER (outside) the timed region, on rank 0 at N = 1:
    #  * the same batch with the reference's constants AS SHIPPED (pixel constants not rescaled from 1182 to this frame size);
    #  * BASELINE configs[1] as the survey restates it: batch 64, the demodulation stage alone (the path's only dense contraction).
    extra = {}
    if rank == 0 and world == 1 and not args.pairs and not args.kernel_tier and not args.no_extras and n == 224:
        other_const = "shipped" if args.constants == "scaled" else "scaled"
        cfg2 = pkg.FtpConfig.as_shipped() if other_const == "shipped" else pkg.FtpConfig.scaled(n)
        sensors2 = [pkg.FtpSensor(ref3, pkg.synth.roi_circle(n), cfg2, cal, neg, fm, max_batch=B, device=dev) for _ in sensors]
        outs2 = [run(s_) for s_ in sensors2]
        torch.cuda.synchronize(dev)
        st2 = Stepper(run, sensors2, outs2, streams, None, torch.cuda.stream)
        k2 = max(10, args.steps // 4)
        el2 = timed_steps(st2, 3, k2, None, lambda: torch.cuda.synchronize(dev), dev)
        extra[f"value_{other_const}_constants"] = {"value": B * k2 / el2, "unit": "frames/s", "steps": k2, "ms_per_step": el2 / k2 * 1e3,
                                                   "frames_with_nonzero_status": int((outs2[0]["status"] != 0).sum().item()),
                                                   "note": f"same batch and sessions in flight, constants {other_const}-{n}; parity of this configuration: tests/test_gpu_parity.py"}
        del sensors2, outs2, st2
        b64 = min(64, B)
        s64 = pkg.FtpSensor(ref3, pkg.synth.roi_circle(n), cfg, cal, neg, fm, max_batch=b64, device=dev)
        f64_ = frames[:b64].contiguous()
        o64 = s64.predict_batch(f64_)
        s64.enable_stage_timing(True)
        dm = 0.0
        for _ in range(reps):
            s64.predict_batch(f64_, o64)
            dm += s64.stage_times_ms().get("pruned-dft demod", 0.0) / reps
        s64.enable_stage_timing(False)
        extra["config1_demod_B64_ms"] = round(dm, 4)
        extra["config1_demod_B64"] = _dft_gemm_record(dm, b64, n, s64, 1)
        del s64, o64

    if rank == 0:
        P = n * n
        dom = max(acc, key=acc.get)
        # algorithmic bytes of the dominant kernel per launch (DESIGN.md "Kernels"):
        #   k_unwrap_fast:        reads wrapped f32 + mask u8 (+ quality f32 for the seed), writes unwrapped f32 -> 9 B/px (13 with the seed search)
        #   k_telea_window_mw:    reads image f32 + bad-mask u8, writes image f32 -> 9 B/px
        #   k_robust_polyfit (x3): reads z f32 + mask u8, writes residual f32 -> 9 B/px per call
        per_px = {"unwrap check (k_unwrap_fast)": 9.0, "inpaint (k_telea_window_mw)": 9.0, "detrend (3x IRLS)": 27.0}.get(dom, 8.0)
        alg_bytes = per_px * P * B
        achieved = alg_bytes / (acc[dom] * 1e-3) / 1e9
        # HBM traffic per launch from the PMC counters: collected by separate rocprofv3 --pmc runs (tools/prof_pmc.sh ->
        # tools/collect_profiles.py -> profiles/traffic_rNN.json), never in this run; reported only when that file was measured on
        # exactly these kernel sources, null otherwise (`traffic_source` says which file and why)
        tj, traffic_source = ({}, None) if args.pairs else _traffic_table()
        traffic = tj.get(dom) if isinstance(tj.get(dom), (int, float)) else None
        # the other two heavy stages, for context: that traffic over this run's stage time
        other = []
        for k in ("unwrap check (k_unwrap_fast)", "detrend (3x IRLS)", "inpaint (k_telea_window_mw)"):
            if k != dom and k in acc and isinstance(tj.get(k), (int, float)) and acc[k] > 0:
                rate = tj[k] / (acc[k] * 1e-3) / 1e9
                other.append({"kernel": k, "kernel_ms": round(acc[k], 4), "traffic": tj[k], "hbm_GBps": round(rate, 1),
                              "frac_of_hbm_peak": round(rate / HBM_PEAK_GBS, 4)})
        line = {
            "metric": "frames/sec (224x224 -> force-map) at batch 256" if not args.pairs else
                      "pairs/sec (uncached (reference, deformed) 224x224 pairs -> force-map), BASELINE configs[4] restated",
            "value": world * B * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[2]: batch {B}/GPU of {n}x{n}x3 fp16 fringe frames ({B} distinct frames) -> {n}x{n} f32 depth map + force scalars, "
                             f"full FTP path (inpaint, demod, unwrap, detrend, compose, force tail), constants {args.constants}-{n}; "
                             f"inputs resident in HBM" if not args.pairs else
                             f"BASELINE configs[4] as SURVEY.md 8(d) restates it: {B} uncached (reference, deformed) pairs/GPU of {n}x{n}x3 fp16 frames, "
                             f"carrier search + reference demodulation per sample (two demodulations per sample, as Code/height_to_force.py:384 "
                             f"runs shape_ftp.main per image; no two-camera / temporal fusion exists upstream), then the full FTP path; "
                             f"constants {args.constants}-{n}; inputs resident in HBM")
                            + ("; one RCCL all-gather of the packed outputs (maps + scalar records + status) per step" if world > 1 else ""),
                "global_batch": world * B, "frame": [n, n, 3], "input_dtype": "fp16", "constants": args.constants,
                "parallelism": f"dp{world}", "inflight_batches": len(sensors), "frames_with_nonzero_status": status_bad,
            },
            "roofline": {
                "bound": "hbm",
                # (the stage names are the library's, fixed per stage: frames beyond the window march's capacity go cluster by cluster)
                "kernel": dom if n * n <= 8 * 14464 else dom.replace("k_telea_window_mw", "k_telea_clusters2 + k_telea_big_clusters"),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel_ms": acc[dom], "algorithmic_bytes_per_launch": alg_bytes,
                "note": {"detrend (3x IRLS)": "three IRLS fits per frame, samples register-resident (one 1024-thread workgroup per frame): ~33 sweeps over "
                                              "the samples per fit (6 normal-equation sweeps, 10 exact medians), bound by VALU issue, not by bytes",
                         }.get(dom, "dominant stage is latency- / issue-bound (dependent steps inside one workgroup per frame), not bandwidth-bound"),
                "other_heavy_stages": other,
            },
            # the one dense contraction of the path: the pruned-DFT stages 1 and 4 as float64 GEMMs on the matrix cores (v_mfma_f64_16x16x4_f64).
            # FLOPs of the two MFMA kernels over the WHOLE demodulation stage's event time (a lower bound on the kernels' own utilisation: the
            # stage also holds the two small non-MFMA stages and the amplitude / atan2 epilogue; per-kernel times: profiles/*_kernel_table.txt)
            "dft_gemm": _dft_gemm_record(acc.get("pruned-dft demod", 0.0), B, n, sensor, 1),
            "stage_ms": {k: round(v, 4) for k, v in acc.items()},
            "serial_ms_per_step": round(sum(acc.values()), 4),
        }
        line.update(extra)
        if cpu_one is not None:
            line["cpu_baseline"] = cpu_one
            line["cpu_baseline_all_cores"] = cpu_all
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
