#!/usr/bin/env python3
"""bench.py — frames/sec of the reference's per-frame camera path (track + EKF update) on MI355X.

One "step" = one camera frame through UpdaterCamera::feed_measurement + try_update (REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:
77-116,139-195) of a running filter, every call through the C-ABI of libplviwo_hip.so:
    plv_tracker_feed_staged   TrackKLT::feed_new_camera / feed_monocular: equalizeHist, 5-level pyramid, FAST top-up detection on the
                              last image (+ cornerSubPix), pyramidal LK, radtan undistortion, 7-point RANSAC, FeatureDatabase update
    plv_vanishing_points + plv_line_tracker_feed
                              TrackLSD::feed_monocular: half-resolution Canny + fast line detector, point-line assignment with the
                              frame's tracked points, line matching, undistortion, classification, LineFeatureDatabase update
    plv_camera_update_points  get_features (pool, unusable measurements, sort, triangulation + refinement, 3 px consistency, cap) ->
                              msckf_update (Jacobians, null space, chi2 gate, compression, EKFUpdate, fp64) -> cleanup_features
    (the driver applies dx to its state, as StateHelper::EKFUpdate does)
    plv_camera_update_lines   get_line_features -> lines_update -> cleanup_lines
The stream is a rendered drive (tests/synth_dataset.py, "street" scene: a camera on a wheeled vehicle going down a corridor with
facades, 200 Hz IMU, 50 Hz wheel odometry); the update consumes the tracker's own database.  Between two frames the driver
(pl-viwo_amd/system.py = SystemManager) feeds the IMU and wheel messages: plv_propagate, plv_cov_clone, plv_cov_marginalize,
plv_wheel_update run for real but outside the timed step, and the next image is staged into HBM (plv_image_stage) there too.
`value` = K / (sum over the K timed steps of the step's wall time, device synchronised at both ends).

Workloads: C (default) = BASELINE configs[2], the configuration the metric is quoted on (752x480, 250 points + lines, 15-clone
window); B = configs[1] (no lines); D = configs[3] (1280x720, 500 points + lines, 20-clone window).

Contract: `python bench.py --gpus N --steps K --warmup W`; N > 1 is launched by torchrun, one rank per GPU.  The path does not shard
("replicas only"): N ranks run N independent replicas of the same stream, `value` is their aggregate frames/s
("weak" scaling), no data-path collective.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

WORKLOADS = {
    # name: (width, height, camera Hz = clone Hz, points, lines, BASELINE config)
    "B": dict(w=752, h=480, hz=15, n_pts=250, lines=False, cfg="configs[1]"),
    "C": dict(w=752, h=480, hz=15, n_pts=250, lines=True, cfg="configs[2]"),
    "D": dict(w=1280, h=720, hz=20, n_pts=500, lines=True, cfg="configs[3]"),
}
PROLOGUE = 24      # frames before the warm-up: initialisation + the first full clone window (untimed set-up)
IMU, WHEEL, CAM = 0, 1, 2


# --------------------------------------------------------------------------------------------------------------- the stream
def build_stream(wl, n_frames, workers):
    """Sensor streams + rendered frames of the drive, in memory.  Rendering forks worker processes: called before anything touches
    the GPU."""
    import synth_dataset as sd
    sd.set_camera(wl["w"], wl["h"])
    sim = sd.simulate(seconds=n_frames / wl["hz"] + 0.2, cam_hz=wl["hz"], style="street")
    tc = sim["cam_times"][:n_frames]
    imgs = sd.render_frames(tc, "street", workers)
    t, wm, am = sim["imu"]
    tw, m1, m2 = sim["wheel"]
    msgs = [(x, IMU, i) for i, x in enumerate(t)] + [(x, WHEEL, i) for i, x in enumerate(tw)] + [(x, CAM, i) for i, x in enumerate(tc)]
    msgs.sort(key=lambda m: (m[0], m[1]))
    return dict(msgs=msgs, imu=np.column_stack([t, wm, am]), wheel=np.column_stack([tw, m1, m2]), cam_t=tc, imgs=imgs, gt=sim["gt"])


def load_options(wl):
    import importlib
    import __graft_entry__ as ge
    import synth_dataset as sd
    ge.load_pkg()
    options = importlib.import_module("plviwo_amd.options")
    d = tempfile.mkdtemp(prefix="plv_bench_cfg_")
    sd.set_camera(wl["w"], wl["h"])
    # the authors' KAIST settings where they apply to one camera: max_msckf 70, sigma_px 1.5, intrinsics calibrated
    # online, polynomial interpolation of order 3 with its covariance; clone rate = camera rate
    op = options.load_options(sd.write_config(d, d, os.path.join(d, "traj.txt"), clone_freq=wl["hz"], n_pts=wl["n_pts"], max_msckf=70,
                                              calib_int=True, sigma_px=1.5))
    op.est.cam.use_lines = bool(wl["lines"])
    return op


class Player:
    """run_bag's message loop, split at the camera messages."""

    def __init__(self, stream, system, staged):
        self.s, self.sys, self.k, self.staged, self.slot = stream, system, 0, staged, 0

    def next_frame(self):
        """feeds IMU / wheel messages up to the next camera message (untimed), stages its image; returns (t, frame index) or None"""
        s, sm = self.s, self.sys
        while self.k < len(s["msgs"]):
            t, kind, i = s["msgs"][self.k]
            self.k += 1
            if kind == IMU:
                r = s["imu"][i]
                sm.feed_measurement_imu(r[0], r[1:4], r[4:7])
            elif kind == WHEEL:
                r = s["wheel"][i]
                sm.feed_measurement_wheel(r[0], r[1], r[2])
            else:
                if self.staged:
                    self.slot ^= 1
                    sm.ctx.image_stage(self.slot, s["imgs"][i])
                    sm.ctx.synchronize()
                return t, i
        return None

    def camera(self, t, i):
        if self.staged:
            self.sys.feed_measurement_camera(t, None, staged_slot=self.slot)
            self.sys.ctx.synchronize()
        else:
            self.sys.feed_measurement_camera(t, self.s["imgs"][i])


def pct(a, q):
    return float(np.percentile(np.asarray(a), q)) if len(a) else None


# --------------------------------------------------------------------------------------------------------------- CPU baseline
class TimedLib:
    """Forwards to a ctypes library and accumulates the wall time spent inside its functions."""
    total = 0.0

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*a):
            t0 = time.perf_counter()
            try:
                return fn(*a)
            finally:
                TimedLib.total += time.perf_counter() - t0
        return call


def cpu_baseline(wl, stream, n_frames, budget_s, threads):
    """The same driver over the CPU oracle (tests/oracle_context.py: fp64 / OpenCV-contract restatement of the same calls, g++ -O3):
    per-frame wall time of feed_measurement_camera on the host, mean / p50 / p99, 1 thread and `threads` threads for the stage the
    reference parallelises (cv::parallel_for_ in calcOpticalFlowPyrLK).  kind = "port": the upstream binary cannot be built (Eigen /
    OpenCV / Boost / ROS absent — DESIGN.md §5).  `inside_oracle` = the part of that time spent inside liboracle.so (the rest is the
    Python bookkeeping of the mirror: database, selection lists)."""
    import importlib
    import __graft_entry__ as ge
    ge.load_pkg()
    import oracle_context as oc
    system = importlib.import_module("plviwo_amd.system")
    out = {}
    for label, nthr in (("1_thread", 1), ("n_threads", threads)):
        if nthr == 1 and label != "1_thread":
            continue
        sm = system.SystemManager(load_options(wl), context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer)
        ctx = sm.ctx
        ctx.lk_threads = nthr
        for o in (ctx.o, ctx.fo, ctx.do, ctx.jo, ctx.lo):
            if not isinstance(o.lib, TimedLib):
                o.lib = TimedLib(o.lib)
        pl = Player(stream, sm, staged=False)
        per, inside, t_begin = [], [], time.perf_counter()
        frames = n_frames if nthr == 1 else max(20, n_frames // 4)
        for f in range(PROLOGUE + frames):
            nf = pl.next_frame()
            if nf is None or (f >= PROLOGUE + 20 and time.perf_counter() - t_begin > budget_s):
                break
            TimedLib.total = 0.0
            t0 = time.perf_counter()
            pl.camera(*nf)
            dt = time.perf_counter() - t0
            if f >= PROLOGUE:
                per.append(dt * 1e3)
                inside.append(TimedLib.total * 1e3)
        st = sm.stats
        out[label] = dict(threads=nthr, frames=len(per), mean_ms=float(np.mean(per)), p50_ms=pct(per, 50), p99_ms=pct(per, 99),
                          inside_oracle_mean_ms=float(np.mean(inside)),
                          split_ms={k.replace("[Time-Cam] ", ""): round(v / max(1, sm.tc.count[k]) * 1e3, 3) for k, v in sm.tc.total.items()
                                    if k.startswith("[Time-Cam]")},
                          cam_features=st["cam_features"], cam_accepted=st["cam_accepted"], lines_accepted=st["lines_accepted"])
        for o in (ctx.o, ctx.fo, ctx.do, ctx.jo, ctx.lo):
            if isinstance(o.lib, TimedLib):
                o.lib = o.lib._lib
    one = out["1_thread"]
    return {"value": 1e3 / one["mean_ms"], "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{one['frames']} frames of the same stream through the same driver over the CPU oracle (liboracle.so, g++ -O3): "
                      f"feed_measurement + try_update {one['mean_ms']:.2f} ms mean / {one['p50_ms']:.2f} p50 / {one['p99_ms']:.2f} p99 per frame on 1 "
                      f"thread, of which {one['inside_oracle_mean_ms']:.2f} ms inside the oracle library; host has {os.cpu_count()} cores",
            "detail": out}


# --------------------------------------------------------------------------------------------------------------- roofline
def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this round (FETCH_SIZE and WRITE_SIZE collected in
    separate runs, profiles/r02/README.md); None when the summary is missing."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02", "bench_*_pmc_hbm.csv")))
    if not found:
        return None, None
    path = found[-1]
    with open(path) as fh:
        next(fh)
        for line in fh:
            k, n, f_kb, w_kb = line.strip().split(",")
            if k.split("<")[0].endswith(kernel):
                return (float(f_kb) + float(w_kb)) * 1024.0, os.path.relpath(path, ROOT) + " (FETCH_SIZE + WRITE_SIZE, KB)"
    return None, None


def reduce_max(elapsed, dist):
    if dist is None:
        return elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="C")
    ap.add_argument("--cpu-frames", type=int, default=200)
    ap.add_argument("--cpu-budget-s", type=float, default=40.0, help="upper bound of the CPU baseline's wall time per pass")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--host-images", action="store_true", help="PCIe-inclusive variant: plv_tracker_feed with the host image")
    ap.add_argument("--render-workers", type=int, default=0, help="0 = min(32, cores)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercises the multi-process plumbing only (tests/test_bench_dist.py)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wl = WORKLOADS[args.workload]

    # ---- the stream first: rendering forks workers, and a fork after the GPU runtime is up is not safe
    stream = None
    if not args.dry_run:
        nprof = max(10, min(40, args.steps))
        n_gpu_frames = PROLOGUE + args.warmup + args.steps + nprof
        n_cpu_frames = 0 if (args.no_cpu or rank != 0) else PROLOGUE + args.cpu_frames
        workers = args.render_workers or max(1, min(32, (os.cpu_count() or 1) // max(1, world)))
        t0 = time.perf_counter()
        stream = build_stream(wl, max(n_gpu_frames, n_cpu_frames), workers)
        t_render = time.perf_counter() - t0

    dist = None
    if world > 1:
        # torch.distributed is plumbing only (barrier + max-reduce of the wall time); the replicas exchange no data.  gloo on CPU
        # tensors: the HIP work is entirely inside libplviwo_hip.so.
        import torch  # noqa: F401
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist = dist_mod
    barrier = (lambda: dist.barrier()) if dist is not None else (lambda: None)

    if args.dry_run:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.05 * (1 + rank))
        elapsed = reduce_max(time.perf_counter() - t0, dist)
        barrier()
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "value": args.steps * world / elapsed, "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "elapsed_s": elapsed,
                              "device_of_rank": local_rank if world > 1 else 0}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import importlib
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    system = importlib.import_module("plviwo_amd.system")
    ndev = max(1, pkg.load_library().plv_device_count())
    device = (local_rank % ndev) if world > 1 else 0   # one GPU per rank; wraps only when a node has fewer GPUs than ranks
    op = load_options(wl)
    sm = system.SystemManager(op, device=device)
    ctx = sm.ctx
    pl = Player(stream, sm, staged=not args.host_images)

    per_frame = {"kept": [], "tracked": []}
    cnt = {k: 0 for k in ("launches", "syncs", "copies", "copy_bytes", "lk_iters", "lines_detected")}

    def sample_counts(f):
        per_frame["tracked"].append(len(ctx.tracker_last()[1]))
        if wl["lines"]:
            per_frame["kept"].append(len(ctx.line_tracker_last()[1]))

    for f in range(PROLOGUE + args.warmup):
        pl.camera(*pl.next_frame())
    if not sm.state.initialized or len(sm.state.clones) < wl["hz"] - 1:
        raise RuntimeError("the filter did not reach a full window during the prologue")
    base = dict(sm.stats)
    tc0 = {k: (sm.tc.total.get(k, 0.0), sm.tc.count.get(k, 0)) for k in list(sm.tc.total)}
    ctx.synchronize()
    barrier()
    elapsed, per = 0.0, []
    for f in range(args.steps):
        nf = pl.next_frame()            # untimed: IMU / wheel messages, cloning, marginalisation, staging of the image
        c0 = pkg.counters()
        t0 = time.perf_counter()
        pl.camera(*nf)                  # timed: feed_measurement + try_update, synchronised
        dt = time.perf_counter() - t0
        elapsed += dt
        per.append(dt * 1e3)
        c1 = pkg.counters()
        for k in cnt:
            cnt[k] += c1[k] - c0[k]
        sample_counts(f)
    ctx.synchronize()
    barrier()
    stats = {k: sm.stats[k] - base.get(k, 0) for k in sm.stats}
    split = {}
    for k, v in sm.tc.total.items():
        if k.startswith("[Time-Cam]"):
            a, c = tc0.get(k, (0.0, 0))
            split[k.replace("[Time-Cam] ", "")] = round((v - a) / max(1, sm.tc.count[k] - c) * 1e3, 4)
    n_state = sm.state.n
    elapsed = reduce_max(elapsed, dist)

    # ---- roofline leg: HIP events around every kernel launch on the ctx stream (a separate pass over the next frames of the stream,
    # so that the event records do not perturb the timed region above)
    roof = None
    if rank == 0:
        import work_model as wm
        ctx.prof_reset()
        b2 = dict(sm.stats)
        done = 0
        for f in range(nprof):
            nf = pl.next_frame()
            if nf is None:
                break
            ctx.prof_enable(True)      # the camera step only: propagation / wheel kernels between the frames are not part of it
            pl.camera(*nf)
            ctx.prof_enable(False)
            done += 1
        table = ctx.prof_table()
        s2 = {k: sm.stats[k] - b2.get(k, 0) for k in sm.stats}
        upd = max(1, s2["cam_updates"])
        F = s2["cam_features"] / max(1, done)
        L = s2["lines_triangulated"] / max(1, done)
        k_cols = n_state - 15 - 6    # every clone + the intrinsics (the IMU pose of the newest frame excluded)
        M = wl["hz"]
        work = wm.frame_work(wl["w"], wl["h"], ctx.pyramid_levels(0), int(np.mean(per_frame["tracked"])), cnt["lk_iters"] / args.steps, 15, F, M, k_cols,
                             n_state, L=L, Ml=max(2, M // 3), kl=k_cols, n_new=max(1, wl["n_pts"] // M),
                             pool_pts=stats["cam_features"] / max(1, args.steps) * 1.5, pool_lines=stats["line_pool"] / max(1, args.steps))
        kernels = {k: v for k, v in table.items() if v[0] > 0}
        name, (n_launch, ms) = max(kernels.items(), key=lambda kv: kv[1][1])
        kind, per_launch = work.get(name, ("hbm", 0.0))
        avg_s = ms / max(n_launch, 1) * 1e-3
        if kind == "hbm":
            achieved, peak, unit = per_launch / avg_s / 1e9, wm.HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = per_launch / avg_s / 1e12, wm.F64_MFMA_PEAK_TF, "TFLOP/s"
        traffic, traffic_src = pmc_traffic(name)
        roof = {"bound": kind, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "traffic": traffic,
                "traffic_source": traffic_src, "kernel": name, "avg_launch_us": avg_s * 1e6, "algorithmic_per_launch": per_launch,
                "launches_per_frame": round(sum(v[0] for v in kernels.values()) / max(1, done), 1),
                "kernel_us_per_frame_total": round(sum(v[1] for v in kernels.values()) / max(1, done) * 1e3, 1),
                "kernels_us_per_frame": {k: round(v[1] / max(1, done) * 1e3, 2) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1][1])},
                "kernels_launches_per_frame": {k: round(v[0] / max(1, done), 2) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1][1])}}
    sm.close()

    cpu = None
    if rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(wl, stream, args.cpu_frames, args.cpu_budget_s, max(1, min(16, os.cpu_count() or 1)))

    if rank == 0:
        mean = lambda a: (round(float(np.mean(a)), 1) if len(a) else None)
        what = f"{wl['w']}x{wl['h']} mono, {wl['n_pts']} KLT points (15x15 window, 5 pyramid levels)"
        if wl["lines"]:
            what += (f" + line front-end (half-resolution Canny + fast line detector: {cnt['lines_detected'] / args.steps:.1f} segments detected, "
                     f"{mean(per_frame['kept'])} kept per frame by the reference's point-line assignment, whose bounding-box test reads the "
                     "end-point coordinates in the wrong order and drops about three of four lines that own a point)")
        line = {
            "metric": ("frames/sec (track+EKF update), 752x480 mono, 250 pts+80 lines; ATE vs CPU ref" if args.workload == "C" else
                       f"frames/sec (track+EKF update), {wl['w']}x{wl['h']} mono, {wl['n_pts']} pts" + (" + lines" if wl["lines"] else "")),
            "value": args.steps * world / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE {wl['cfg']}: {what}; {wl['hz']}-clone window ({wl['hz']} Hz camera and clones, 1 s), n = {n_state}; "
                            "rendered street-corridor drive with IMU + wheel odometry; the update consumes the tracker's own database",
                "step": "plv_camera_frame = plv_tracker_feed_staged -> plv_vanishing_points + plv_line_tracker_feed -> plv_camera_try_update "
                        "(plv_camera_update_points -> dx applied -> plv_camera_update_lines -> dx applied); sequential, device synchronised at both "
                        "ends of every step; IMU propagation, cloning, marginalisation, wheel updates and image staging run between the steps, "
                        "untimed",
                "replicas": world, "n_state": n_state,
                "latency_ms": {"mean": float(np.mean(per)), "p50": pct(per, 50), "p99": pct(per, 99), "max": float(np.max(per))},
                "host_split_ms_per_frame": split,
                "per_frame": {"tracked_points": mean(per_frame["tracked"]), "lines_detected": round(cnt["lines_detected"] / args.steps, 1),
                              "lines_kept": mean(per_frame["kept"]),
                              "msckf_features": round(stats["cam_features"] / args.steps, 2),
                              "msckf_accepted": round(stats["cam_accepted"] / args.steps, 2),
                              "line_pool": round(stats["line_pool"] / args.steps, 2),
                              "lines_triangulated": round(stats["lines_triangulated"] / args.steps, 2),
                              "lines_accepted": round(stats["lines_accepted"] / args.steps, 3)},
                "submissions_per_frame": {"kernel_launches": round(cnt["launches"] / args.steps, 1), "host_synchronisations": round(cnt["syncs"] / args.steps, 1),
                                          "copies": round(cnt["copies"] / args.steps, 1), "copy_kB": round(cnt["copy_bytes"] / args.steps / 1024, 1),
                                          "lk_iterations": round(cnt["lk_iters"] / args.steps)},
                "updates": {"point_updates": stats["cam_updates"], "line_updates": stats["line_updates"], "not_psd": stats["not_psd"],
                            "frames": args.steps},
                "images": "host (PCIe inside the step)" if args.host_images else "resident in HBM (plv_image_stage between the steps)",
                "front_end_arithmetic": "u8/int16/int64 exact + f32 2x2 solve", "update_arithmetic": "f64",
                "render_s": round(t_render, 1),
            },
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
