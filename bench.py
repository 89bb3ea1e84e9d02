#!/usr/bin/env python3
"""bench.py — frames/sec of the reference's per-frame camera path (track + EKF update) on MI355X.

One "step" = one camera frame through UpdaterCamera::feed_measurement + try_update (REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:
77-116,139-195) of a running filter, one call through the C-ABI of libplviwo_hip.so (plv_camera_frame), which runs
    plv_tracker_feed[_staged] TrackKLT::feed_new_camera / feed_monocular: equalizeHist, 5-level pyramid, FAST top-up detection on the
                              last image (+ cornerSubPix), pyramidal LK, radtan undistortion, 7-point RANSAC, FeatureDatabase update
    plv_vanishing_points + plv_line_tracker_feed
                              TrackLSD::feed_monocular: half-resolution Canny + fast line detector, point-line assignment with the
                              frame's tracked points, line matching, undistortion, classification, LineFeatureDatabase update
    plv_camera_try_update     get_features (pool, unusable measurements, sort, triangulation + refinement, 3 px consistency, cap) ->
                              get_line_features' state recorded -> msckf_update (Jacobians, null space, chi2 gate, compression,
                              EKFUpdate, fp64) -> cleanup_features -> dx applied -> get_line_features + lines_update ->
                              cleanup_lines -> dx applied
The stream is a rendered drive (tests/synth_dataset.py; workload C: the "boulevard" scene, a side-looking camera on a wheeled vehicle
driving along a facade of slanted strip courses, on which TrackLSD keeps the metric's 80 lines per frame; B / D / C_avenue: the
"avenue" corridor of rounds 3-4; 200 Hz IMU, 50 Hz wheel odometry); the update consumes the tracker's own database.  Between two frames the driver
(pl-viwo_amd/system.py = SystemManager) feeds the IMU and wheel messages: plv_propagate, plv_cov_clone, plv_cov_marginalize,
plv_wheel_update run for real but outside the timed step.

Two timed segments of K steps each over consecutive frames of the stream: (1) the image already resident in HBM when the step starts
(plv_image_stage between the steps) -> `value`; (2) the image handed over as a host buffer, its PCIe copy inside the step ->
`config.pcie_inclusive`.  Every step ends with plv_ctx_synchronize (ctx stream, the detection side stream, the line worker).
`value` = K / (sum over the K timed steps of the step's wall time).

`cpu_baseline`: the same frames through oracle/frame_oracle.cpp (the CPU frame compiled end to end, g++ -O3), timed inside the
library with std::chrono::steady_clock.  `config.stress`: the update chain alone at SURVEY §8(d)'s sizing (F = 70 features x 15
observations, then L = 80 lines x 15) on the HIP library and on the oracle.

Workloads: C (default) = BASELINE configs[2], the configuration the metric is quoted on (752x480, 250 tracked points + lines,
15-clone window); B = configs[1] (no lines); D = configs[3] (1280x720, 500 points + lines, 20-clone window).

Contract: `python bench.py --gpus N --steps K --warmup W`; N > 1 is launched by torchrun, one rank per GPU.  The path does not shard
("replicas only"): N ranks run N independent replicas of the same stream, `value` is their aggregate frames/s
("weak" scaling), no data-path collective.  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import glob
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
    # points = the tracked-point count the metric names; num_features = TrackKLT's n_pts setting that sustains it on this scene (the
    # reference tops a grid cell up only while it holds fewer than half its share, TrackKLT.cpp:474-495, so the count it sustains sits
    # at ~70 % of the setting)
    # scene / mount: tests/synth_dataset.py.  "boulevard" (round 5) = a side-looking camera along a facade of slanted strip courses:
    # the scene on which TrackLSD keeps the metric's 80 lines per frame (see _mips_boulevard for why 'avenue' keeps 30); "avenue" =
    # the forward-looking corridor drive of rounds 3-4 (workload C_avenue keeps it as a variant)
    "B": dict(w=752, h=480, hz=15, points=250, num_features=360, lines=False, cfg="configs[1]", scene="avenue", mount=(12.0, 0.0)),
    "C": dict(w=752, h=480, hz=15, points=250, num_features=440, lines=True, cfg="configs[2]", scene="boulevard", mount=(16.0, 90.0)),
    "C_avenue": dict(w=752, h=480, hz=15, points=250, num_features=360, lines=True, cfg="configs[2]", scene="avenue", mount=(12.0, 0.0)),
    "D": dict(w=1280, h=720, hz=20, points=500, num_features=780, lines=True, cfg="configs[3]", scene="avenue", mount=(12.0, 0.0)),
}
PROLOGUE = 60      # frames before the warm-up: initialisation, the first full clone window, and the library's buffers and threads reaching
                   # their steady state (untimed set-up: with 24 the driver's 20-step run sat 10-20 % above the 200-step one)
LEAD_IN = 2        # untimed steps at the start of every timed segment, after its garbage collection (see timed_segment)
IMU, WHEEL, CAM = 0, 1, 2
ROUND = "r06"


# --------------------------------------------------------------------------------------------------------------- the stream
def build_stream(wl, n_frames, workers, cache=None):
    """Sensor streams + rendered frames of the drive, in memory.  Rendering forks worker processes: called before anything touches
    the GPU — or not at all when `cache` names a stream an earlier run saved (a profiled run must not fork: the profiler's library
    has initialised the GPU before Python starts)."""
    import synth_dataset as sd
    sd.set_camera(wl["w"], wl["h"])
    sd.set_mount(*wl["mount"])
    SCENE = wl["scene"]
    sim = sd.simulate(seconds=n_frames / wl["hz"] + 0.2, cam_hz=wl["hz"], style=SCENE)
    tc = sim["cam_times"][:n_frames]
    imgs = None
    if cache and os.path.exists(cache):
        z = np.load(cache)
        if z["imgs"].shape[0] >= n_frames and z["imgs"].shape[1:] == (wl["h"], wl["w"]) and np.array_equal(z["tc"][:n_frames], tc):
            imgs = list(z["imgs"][:n_frames])
    if imgs is None:
        imgs = sd.render_frames(tc, SCENE, workers)
        if cache:
            np.savez(cache, imgs=np.array(imgs), tc=tc)
    t, wm, am = sim["imu"]
    tw, m1, m2 = sim["wheel"]
    msgs = [(x, IMU, i) for i, x in enumerate(t)] + [(x, WHEEL, i) for i, x in enumerate(tw)] + [(x, CAM, i) for i, x in enumerate(tc)]
    msgs.sort(key=lambda m: (m[0], m[1]))
    return dict(msgs=msgs, imu=np.column_stack([t, wm, am]), wheel=np.column_stack([tw, m1, m2]), cam_t=tc, imgs=imgs, gt=sim["gt"])


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def pin_to_device(pkg, device, mode, rank_on_node=None):
    """The process (and every thread it starts afterwards) onto the cores next to the GPU: its NUMA node's, or one L3 complex of that
    node.  What `numactl --cpunodebind` does for a launcher; a two-socket host otherwise schedules the caller and the library's line
    threads on either socket, and a frame's ~40 hand-overs between them and the device's pinned result blocks cross the sockets.
    Returns what was done, for the bench line."""
    if mode == "none" or not hasattr(os, "sched_setaffinity"):
        return {"mode": "none"}
    try:
        node = pkg.device_numa_node(device)
        allowed = os.sched_getaffinity(0)
        if node < 0:
            return {"mode": "none", "why": "the device's NUMA node is unknown"}
        cpus = [c for c in _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) if c in allowed]
        if len(cpus) < 4:
            return {"mode": "none", "why": "fewer than four allowed cores on the device's node"}
        chosen, info = cpus, {"mode": "node", "numa_node": node, "cpus": len(cpus)}
        if mode in ("ccx", "ccx2"):
            groups = {}
            for c in cpus:
                try:
                    key = open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list").read().strip()
                except OSError:
                    key = "all"
                groups.setdefault(key, []).append(c)
            groups = [g for g in groups.values() if len(g) >= 8]
            if groups:
                def busy_snapshot():
                    out = {}
                    for line in open("/proc/stat"):
                        if line.startswith("cpu") and line[3].isdigit():
                            f = line.split()
                            v = [int(x) for x in f[1:9]]
                            out[int(f[0][3:])] = (sum(v) - v[3] - v[4], sum(v))
                    return out
                if rank_on_node is None:
                    # three 0.1 s samples; a complex is as busy as its busiest sample (other tenants' load comes in bursts, and a
                    # burst on the pinned cores costs whole time slices: frames of several ms)
                    load = [0.0] * len(groups)
                    a = busy_snapshot()
                    for _ in range(3):
                        time.sleep(0.1)
                        b = busy_snapshot()
                        for gi, g in enumerate(groups):
                            load[gi] = max(load[gi], sum(b[c][0] - a[c][0] for c in g) / max(1, sum(b[c][1] - a[c][1] for c in g)))
                        a = b
                    k = int(np.argmin(load))
                    info["ccx_busy_fraction"] = round(float(load[k]), 3)
                else:
                    k = rank_on_node % len(groups)
                chosen = groups[k]
                if len(chosen) < 10 and len(groups) > 1:   # (no SMT: the caller + worker + seven helpers need more than one complex's cores)
                    chosen = chosen + groups[k + 1 if k + 1 < len(groups) else k - 1]
                if mode == "ccx2" and len(groups) > 1:    # (measurement aid: two adjacent L3 complexes — room for more helper threads)
                    chosen = chosen + groups[k + 1 if k + 1 < len(groups) else k - 1]
                info.update(mode=mode, cpus=len(chosen), first_cpu=min(chosen))
        os.sched_setaffinity(0, set(chosen))
        return info
    except OSError as e:
        return {"mode": "none", "why": str(e)}


def load_options(wl):
    import importlib
    import __graft_entry__ as ge
    import synth_dataset as sd
    ge.load_pkg()
    options = importlib.import_module("plviwo_amd.options")
    d = tempfile.mkdtemp(prefix="plv_bench_cfg_")
    sd.set_camera(wl["w"], wl["h"])
    sd.set_mount(*wl["mount"])
    # the authors' KAIST settings where they apply to one camera: max_msckf 70, sigma_px 1.5, intrinsics calibrated
    # online, polynomial interpolation of order 3 with its covariance; clone rate = camera rate
    op = options.load_options(sd.write_config(d, d, os.path.join(d, "traj.txt"), clone_freq=wl["hz"], n_pts=wl["num_features"], max_msckf=70,
                                              calib_int=True, sigma_px=1.5))
    op.est.cam.use_lines = bool(wl["lines"])
    return op


class Player:
    """run_bag's message loop, split at the camera messages."""

    def __init__(self, stream, system, staged, pinned=False):
        # staged: the image is put into HBM between the steps (plv_image_stage); pinned (with staged off): the image is written into
        # one of the library's page-locked blocks between the steps (plv_image_buffer: where the ROS callback's copy of the message
        # would land) and crosses PCIe inside the step; neither: any host buffer (the call copies it into such a block first)
        self.s, self.sys, self.k, self.staged, self.slot, self.pinned, self.pin_view = stream, system, 0, staged, 0, pinned, None

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
                elif self.pinned:
                    self.slot ^= 1
                    self.pin_view = sm.ctx.image_buffer(self.slot)
                    np.copyto(self.pin_view, s["imgs"][i])
                return t, i
        return None

    def image(self, i):
        return self.pin_view if (self.pinned and not self.staged) else self.s["imgs"][i]

    def camera(self, t, i):
        """the whole frame through the Python driver (prologue, warm-up, the CPU frame, `ms_per_step_python`)"""
        if self.staged:
            self.sys.feed_measurement_camera(t, None, staged_slot=self.slot)
        else:
            self.sys.feed_measurement_camera(t, self.image(i))
        if hasattr(self.sys.ctx, "synchronize"):
            self.sys.ctx.synchronize()     # ctx stream + detection side stream + line worker

    def camera_timed(self, t, i):
        """The step as the reference's C++ caller makes it (UpdaterCamera.cpp:77-195 has no marshalling layer): the driver's frame
        bookkeeping and the call's arguments are put into their C structures first (SystemManager.camera_prepare, untimed, like the IMU /
        wheel work between the frames), the timed region is plv_camera_frame + plv_ctx_synchronize and nothing else (two ctypes calls
        on prepared arguments), the results are counted afterwards.  Returns the step's seconds."""
        sm = self.sys
        prep = sm.camera_prepare(t, None if self.staged else self.image(i), None, self.slot if self.staged else None)
        if prep is None:
            t0 = time.perf_counter()
            self.camera(t, i)
            return time.perf_counter() - t0
        t0 = time.perf_counter()
        sm.camera_run(prep, sync=True)
        dt = time.perf_counter() - t0
        # (the reference's TimeChecker label of this region, UpdaterCamera.cpp:79,190: recorded from the measured time, not around it)
        sm.tc.total["[Time-Cam] feed measurement + try_update"] = sm.tc.total.get("[Time-Cam] feed measurement + try_update", 0.0) + dt
        sm.tc.count["[Time-Cam] feed measurement + try_update"] = sm.tc.count.get("[Time-Cam] feed measurement + try_update", 0) + 1
        sm.camera_finish(prep)
        return dt


def cgroup_cpu():
    """cpu.stat of this process's control group (v2), or None"""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as fh:
            kv = dict(line.split() for line in fh if line.strip())
        return {k: int(kv.get(k, 0)) for k in ("usage_usec", "nr_periods", "nr_throttled")}
    except (OSError, ValueError):
        return None


def pct(a, q):
    return float(np.percentile(np.asarray(a), q)) if len(a) else None


# --------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(wl, stream, n_frames, budget_s, thread_counts):
    """The same frames through the compiled CPU frame (oracle/frame_oracle.cpp behind tests/oracle_context.py: the reference's
    feed_measurement + try_update restated in C++ fp64 / OpenCV-contract arithmetic, g++ -O3): per-frame time measured INSIDE the
    library with std::chrono::steady_clock (`value`), and the wall time of the driver's call around it.  1 thread, and more threads
    for the one stage the reference parallelises (cv::parallel_for_ in calcOpticalFlowPyrLK).  kind = "port": the upstream binary
    cannot be built (Eigen / OpenCV / Boost / ROS absent — DESIGN.md §5)."""
    import importlib
    import __graft_entry__ as ge
    ge.load_pkg()
    import oracle_context as oc
    import oracle_lib
    system = importlib.import_module("plviwo_amd.system")
    out = {}
    #  a second build of the same sources for this machine (-O3 -march=native, contraction on), timing only
    native = oracle_lib.build_native(tempfile.mkdtemp(prefix="plv_oracle_native_"))
    passes = [(n, None) for n in thread_counts] + ([(thread_counts[0], native), (thread_counts[-1], native)] if native else [])
    for nthr, lib_path in passes:
        oracle_lib.FRAME_LIB = lib_path
        sm = system.SystemManager(load_options(wl), context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer)
        oracle_lib.FRAME_LIB = None
        ctx = sm.ctx
        ctx.lk_threads = nthr
        pl = Player(stream, sm, staged=False)
        per, inside, parts, t_begin = [], [], [], time.perf_counter()
        frames = n_frames if (nthr == thread_counts[0] and lib_path is None) else max(20, n_frames // 4)
        tracked, kept = [], []
        for f in range(PROLOGUE + frames):
            nf = pl.next_frame()
            if nf is None or (f >= PROLOGUE + 20 and time.perf_counter() - t_begin > budget_s):
                break
            tm0 = ctx.frame.timing_ms.copy()
            t0 = time.perf_counter()
            pl.camera(*nf)
            dt = time.perf_counter() - t0
            if f >= PROLOGUE:
                d = ctx.frame.timing_ms - tm0
                per.append(dt * 1e3)
                inside.append(d[5])
                parts.append(d[:5])
                tracked.append(len(ctx.tracker_last()[1]))
                kept.append(len(ctx.line_tracker_last()[1]))
        st = sm.stats
        parts = np.mean(parts, axis=0)
        out[f"{nthr}_thread" + ("s" if nthr > 1 else "") + ("_native" if lib_path else "")] = dict(
            threads=nthr, build="-O3 -march=native -ffp-contract=fast" if lib_path else "-O3 -march=x86-64-v3 -ffp-contract=off (oracle/Makefile, the parity build)", frames=len(per), inside_mean_ms=float(np.mean(inside)), inside_p50_ms=pct(inside, 50), inside_p99_ms=pct(inside, 99),
            driver_call_mean_ms=float(np.mean(per)),
            split_ms={"feed points (equalize, pyramid, detection, LK, RANSAC, database)": round(parts[0], 3), "feed lines": round(parts[1], 3),
                      "get_features + msckf_update + cleanup": round(parts[2], 3), "get_line_features": round(parts[3], 3),
                      "lines_update + cleanup": round(parts[4], 3)},
            tracked_points=round(float(np.mean(tracked)), 1), lines_kept=round(float(np.mean(kept)), 1),
            cam_features=st["cam_features"], cam_accepted=st["cam_accepted"], lines_accepted=st["lines_accepted"])
        sm.close()
    one = out["1_thread"]
    best_key = min(out, key=lambda kk: out[kk]["inside_mean_ms"])
    best = out[best_key]
    nat1 = out.get("1_thread_native")
    return {"value": 1e3 / one["inside_mean_ms"], "unit": "frames/s", "cores": 1, "kind": "port",
            "value_native_build": (1e3 / nat1["inside_mean_ms"]) if nat1 else None, "ms_per_frame_1_thread_native_build": nat1["inside_mean_ms"] if nat1 else None,
            #  the fastest thread count next to the 1-thread figure, each with its cores
            "value_all_cores": 1e3 / best["inside_mean_ms"], "cores_all_cores": best["threads"], "ms_per_frame_1_thread": one["inside_mean_ms"],
            "ms_per_frame_all_cores": best["inside_mean_ms"], "host_cores": os.cpu_count(),
            "threads_tried": sorted(set(d["threads"] for d in out.values())),
            "sample": f"{one['frames']} frames of the same stream through oracle/frame_oracle.cpp (feed_measurement + try_update compiled end to "
                      f"end, g++ -O3, timed inside the library with steady_clock): {one['inside_mean_ms']:.2f} ms mean / {one['inside_p50_ms']:.2f} p50 / "
                      f"{one['inside_p99_ms']:.2f} p99 per frame on 1 thread ({one['driver_call_mean_ms']:.2f} ms with the Python driver's call around "
                      f"it); host has {os.cpu_count()} cores.  Resources side by side: the HIP path = 1 GPU + its host threads (caller, line worker, "
                      f"polling helpers: see host_threads / host_cpu_seconds_per_second at the top level) pinned to one L3 complex; this baseline = "
                      f"{best['threads']} CPU thread(s) at its best ({best_key}), every allowed CPU available to it",
            "detail": out}


# --------------------------------------------------------------------------------------------------------------- stress leg
def stress_update(pkg, device, steps, with_cpu):
    """SURVEY §8(d)'s update sizing on its own: F = 70 MSCKF features x 15 observations (k = 98 of n = 113), then L = 80 lines x 15
    observations, each as Jacobians + null space + gate + compression + EKFUpdate on the resident covariance (restored every step so
    that every step does the same work); the same two updates on the CPU oracle."""
    import bench_chain as bc
    import synth
    cfg = pkg.default_config(752, 480)
    cfg.device = device
    ctx = pkg.Context(cfg)
    scene = synth.vio_scene(n_clones=15, F=bc.F_FEATS, M=bc.M_OBS, seed=3, noise_px=0.4)
    st, tr = synth.scene_views(pkg, scene)
    cols = ctx.jacobian_columns(st, tr)
    ls = synth.line_scene(scene, L=bc.N_LINES, M=bc.M_OBS, noise_px=0.4)
    lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=ls["lines"])
    cols_l = ctx.line_jacobian_columns(st, lt)
    P = synth.spd_cov(scene["n_state"])
    ctx.cov_upload(P)
    ctx.cov_checkpoint()
    n = scene["n_state"]
    acc = [0, 0]

    def one():
        ctx.cov_rollback()
        ctx.build_jacobians_resident(st, tr, cols, 2 * bc.M_OBS)
        rc, dx, a, nr = ctx.msckf_update_resident(n, bc.SIGMA2)
        acc[0] = int(a.sum())
        ctx.build_line_jacobians_resident(st, lt, cols_l, bc.LINE_LD)
        rc2, dx2, a2, nr2 = ctx.msckf_update_resident(n, bc.SIGMA2, res_norm_gate=0.0)
        acc[1] = int(a2.sum())
        if rc != 0 or rc2 != 0:
            raise RuntimeError("stress update rejected")

    for _ in range(5):
        one()
    ctx.synchronize()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        one()
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(10):
        one()
    ctx.prof_enable(False)
    table = {k: v for k, v in ctx.prof_table().items() if v[0] > 0}
    ctx.close()
    out = {"what": f"update chain alone: F = {bc.F_FEATS} features x {bc.M_OBS} observations (k = {len(cols)}, n = {n}), then L = {bc.N_LINES} lines x "
                   f"{bc.M_OBS} (k = {len(cols_l)}): Jacobians + null space + chi2 gate + compression + EKFUpdate each, covariance restored per step",
           "ms_per_step": float(np.mean(ts)), "p50_ms": pct(ts, 50), "steps": steps, "accepted_features": acc[0], "accepted_lines": acc[1],
           "kernels_us_per_step": {k: round(v[1] / 10 * 1e3, 2) for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])}}
    if with_cpu:
        import oracle_lib
        orc, jo = oracle_lib.load(), oracle_lib.load_jac(pkg)
        q95 = synth.q95_table()
        cs = []
        for _ in range(3):
            t0 = time.perf_counter()
            rows, Hf, Hx, res = jo.build_jacobians(st, tr, jo.columns(st, tr), 2 * bc.M_OBS)
            rc, P1, _, _, _ = orc.msckf_update(P, rows, Hf, Hx, res, jo.columns(st, tr), bc.SIGMA2, q95)
            rows, Hf, Hx, res = jo.build_line_jacobians(st, lt, jo.line_columns(st, lt), bc.LINE_LD)
            orc.msckf_update(P1, rows, Hf, Hx, res, jo.line_columns(st, lt), bc.SIGMA2, q95, res_norm_gate=0.0)
            cs.append((time.perf_counter() - t0) * 1e3)
        out["cpu_oracle_ms_per_step"] = float(np.min(cs))
        out["ratio_vs_cpu_oracle"] = out["cpu_oracle_ms_per_step"] / out["ms_per_step"]
    return out


# --------------------------------------------------------------------------------------------------------------- roofline
def pmc_table():
    """HBM bytes per launch per kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs,
    profiles/rNN/README.md): the newest summary of this workload family; {} when none is committed."""
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*", "bench_c_pmc_hbm.csv")))
    if not found:
        return {}, None
    path = found[-1]
    out = {}
    with open(path) as fh:
        next(fh)
        for line in fh:
            k, n, f_kb, w_kb = line.strip().split(",")
            k = k.split("<")[0]
            if k and f_kb != "nan" and w_kb != "nan":
                out[k] = (float(f_kb) + float(w_kb)) * 1024.0
    return out, os.path.relpath(path, ROOT) + " (FETCH_SIZE + WRITE_SIZE, KB per dispatch)"


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
    ap.add_argument("--no-stress", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the timed segments with host images")
    ap.add_argument("--images", choices=("resident", "host", "pinned"), default="resident",
                    help="measurement aid: where the FIRST timed segment's images are when a step starts (resident = in HBM, the contract's "
                         "`value`; host = any host buffer; pinned = the library's page-locked block, plv_image_buffer)")
    ap.add_argument("--no-variants", action="store_true", help="skip the short segments with non-default library settings (config.variants)")
    ap.add_argument("--render-workers", type=int, default=0, help="0 = min(32, cores)")
    ap.add_argument("--stream-cache", default=None, help=".npz of the rendered frames: written when missing, loaded (no rendering, no fork) when present")
    ap.add_argument("--alternate-modes", default=None, help="measurement aid: e.g. 0,1 — the timed segment cycles through these "
                    "plv_update_compression_mode settings (0, 1) frame by frame and stderr gets the mean step time of each (drift-free A/B)")
    ap.add_argument("--alternate-spin", default=None, help="measurement aid: e.g. 300,0 — plv_line_worker_config polling budgets (us) cycled frame by frame")
    ap.add_argument("--alternate-fit", default=None, help="measurement aid: e.g. 2,0 — segment-fitter thread counts cycled frame by frame")
    ap.add_argument("--alternate-images", default=None, help="measurement aid: e.g. resident,pinned,host — the hand-over of the image cycled frame by "
                    "frame in the first timed segment, mean step time of each on stderr")
    ap.add_argument("--alternate-knobs", default=None, help="measurement aid: e.g. 0,1 — plv_debug_knobs masks cycled frame by frame, "
                    "mean step time of each on stderr")
    ap.add_argument("--pin", choices=("none", "node", "ccx", "ccx2"), default=os.environ.get("PLV_BENCH_PIN", "ccx"),
                    help="CPU affinity of this process and the library's threads: the cores of the GPU's NUMA node, or one L3 complex of that "
                         "node (the least busy one; with several ranks, one complex per rank)")
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
        seg2 = 0 if args.no_pcie else 3 * args.steps      # (one segment alternating resident / any host buffer / pinned block)
        nvar = 0 if args.no_variants else max(10, min(40, args.steps))
        npy = max(10, min(40, args.steps))      # (the segment timed through the Python driver: config.ms_per_step_python)
        n_gpu_frames = PROLOGUE + args.warmup + args.steps + seg2 + npy + nprof + 6 * nvar + 8 * LEAD_IN   # (three alternating variants of 2 * nvar frames)
        n_cpu_frames = 0 if (args.no_cpu or rank != 0) else PROLOGUE + args.cpu_frames
        workers = args.render_workers or max(1, min(32, (os.cpu_count() or 1) // max(1, world)))
        t0 = time.perf_counter()
        stream = build_stream(wl, max(n_gpu_frames, n_cpu_frames), workers, args.stream_cache)
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
    cpus_before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = pin_to_device(pkg, device, args.pin, local_rank if world > 1 else None)   # (before the library starts its threads: they inherit it)
    op = load_options(wl)
    sm = system.SystemManager(op, device=device)
    ctx = sm.ctx
    pl = Player(stream, sm, staged=True)

    alt_modes = [int(m) for m in args.alternate_modes.split(",")] if args.alternate_modes else None
    alt_knobs = [int(m) for m in args.alternate_knobs.split(",")] if args.alternate_knobs else None
    alt_spin = [int(m) for m in args.alternate_spin.split(",")] if args.alternate_spin else None
    alt_fit = [int(m) for m in args.alternate_fit.split(",")] if args.alternate_fit else None
    alt_img = args.alternate_images.split(",") if args.alternate_images else None

    def timed_segment(n_steps, hook=None, through_python=False, images=None):
        ai = images or alt_img    # the image's hand-over alternating frame by frame ("resident", "host", "pinned")
        per_frame = {"kept": [], "tracked": []}
        cnt = {k: 0 for k in ("launches", "syncs", "copies", "copy_bytes", "lk_iters", "lines_detected", "frame_ns", "sync_ns", "ambiguous_frames",
                              "redone_frames", "whitened_frames")}
        ctx.synchronize()
        gc.collect()
        gc.disable()      # (the driver is Python: its collector must not land inside a 0.6 ms step)
        # the collection above is a pause of milliseconds in which the library's threads go to sleep and the caches cool: two untimed
        # steps (more warm-up, same code path) lead back into the steady state the timed steps are meant to show
        for li in range(LEAD_IN):
            if ai:
                pl.staged, pl.pinned = ai[(li - LEAD_IN) % len(ai)] == "resident", ai[(li - LEAD_IN) % len(ai)] == "pinned"
            nf = pl.next_frame()
            if hook:
                hook(0)
            pl.camera(*nf)
        base = dict(sm.stats)
        tc0 = {k: (sm.tc.total.get(k, 0.0), sm.tc.count.get(k, 0)) for k in list(sm.tc.total)}
        ctx.synchronize()
        barrier()
        cg0, wall0 = cgroup_cpu(), time.perf_counter()
        elapsed, per, grew, slow = 0.0, [], [], []
        trace_frames = bool(os.environ.get("PLV_BENCH_FRAMES"))
        fw = None
        if os.environ.get("PLV_BENCH_FAULTWHERE"):     # tools/ubench/faultwhere.c: address + ip of every minor fault of this thread
            import ctypes
            fw = ctypes.CDLL(os.environ["PLV_BENCH_FAULTWHERE"])
            if fw.fw_start() != 0:
                fw = None
        chain0, routes0 = pkg.chain_count(), pkg.route_counts()
        last_stats = dict(sm.stats)
        for f in range(n_steps):
            if ai:
                pl.staged, pl.pinned = ai[f % len(ai)] == "resident", ai[f % len(ai)] == "pinned"
            nf = pl.next_frame()            # untimed: IMU / wheel messages, cloning, marginalisation (+ staging of the image in segment 1)
            if nf is None:
                raise RuntimeError(f"bench: the rendered stream ended {n_steps - f} frames before a timed segment did (n_gpu_frames undersized)")
            if hook:
                hook(f)
            if alt_modes:
                ctx.update_compression_mode(alt_modes[f % len(alt_modes)])
            if alt_knobs:
                pkg.debug_knobs(alt_knobs[f % len(alt_knobs)])
            if alt_spin:
                pkg.line_worker_config(alt_spin[f % len(alt_spin)], -1)
            if alt_fit:
                pkg.line_worker_config(-1, alt_fit[f % len(alt_fit)])
            c0 = pkg.counters()
            a0 = pkg.alloc_count()
            cls0 = (pkg.chain_count(), pkg.speculation_counts()) if trace_frames else None
            ph0 = pkg.phase_counters() if trace_frames else None
            if fw:
                import resource
                ru0 = resource.getrusage(resource.RUSAGE_THREAD)
                fw.fw_mark()
            if through_python:              # (config.ms_per_step_python: the marshalling of the Python driver inside the step, as rounds 1-4 timed it)
                t0 = time.perf_counter()
                pl.camera(*nf)
                dt = time.perf_counter() - t0
            else:                           # timed: plv_camera_frame + plv_ctx_synchronize; everything the frame started has finished at return
                dt = pl.camera_timed(*nf)
            if fw:
                fw.fw_end()
                ru1 = resource.getrusage(resource.RUSAGE_THREAD)
                fw_ru = fw_ru + [(ru1.ru_minflt - ru0.ru_minflt, (ru1.ru_stime - ru0.ru_stime) * 1e6, (ru1.ru_utime - ru0.ru_utime) * 1e6,
                                  ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw)] if f else []
            elapsed += dt
            per.append(dt * 1e3)
            if pkg.alloc_count() != a0:
                grew.append(f)
            if trace_frames and dt > 1e-3:      # where did a slow step spend its time?
                ph1 = pkg.phase_counters()
                c1_ = pkg.counters()
                slow.append((f, round(dt * 1e3, 3), {k: round((ph1[k] - ph0[k]) * 1e-6, 3) for k in ph1},
                             round((c1_["frame_ns"] - c0["frame_ns"]) * 1e-6, 3), round((c1_["sync_ns"] - c0["sync_ns"]) * 1e-6, 3)))
            c1 = pkg.counters()
            for k in c1:
                cnt[k] += c1[k] - c0[k]
            if trace_frames:      # what this frame's updates worked on, next to its time (PLV_BENCH_FRAMES=1)
                cur = dict(sm.stats)
                ph1t = pkg.phase_counters()
                per_frame.setdefault("trace", []).append((round(dt * 1e3, 3), cur["cam_features"] - last_stats.get("cam_features", 0), cur["line_pool"] - last_stats.get("line_pool", 0),
                                                          cur["lines_triangulated"] - last_stats.get("lines_triangulated", 0), cur["lines_accepted"] - last_stats.get("lines_accepted", 0),
                                                          c1["lk_iters"] - c0["lk_iters"], c1["lines_detected"] - c0["lines_detected"]) +
                                                         tuple(round((ph1t[k] - ph0[k]) * 1e-3, 1) for k in ("flow_wait", "points", "lines", "w_maps", "w_extract", "w_feed")))
                sp1 = pkg.speculation_counts()
                per_frame.setdefault("classes", []).append((dt * 1e3, pkg.chain_count() - cls0[0], sp1[0] - cls0[1][0], sp1[2] - cls0[1][2] + sp1[3] - cls0[1][3],
                                                            cur["lines_accepted"] - last_stats.get("lines_accepted", 0), c1["launches"] - c0["launches"]))
                last_stats = cur
            per_frame["tracked"].append(len(ctx.tracker_last()[1]))
            if wl["lines"]:
                per_frame["kept"].append(len(ctx.line_tracker_last()[1]))
            _, route, amb = ctx.update_compression_mode()
            cnt["ambiguous_frames"] += 1 if amb > 0 else 0
            cnt["redone_frames"] += 1 if route == 3 else 0
            cnt["whitened_frames"] += 1 if route == 4 else 0
        ctx.synchronize()
        cnt["chained"] = pkg.chain_count() - chain0
        cnt["routes"] = [a - b for a, b in zip(pkg.route_counts(), routes0)]
        if fw:
            fw.fw_stop(os.environ.get("PLV_BENCH_FAULTWHERE_OUT", "").encode())
            m = np.mean(np.array(fw_ru, float), axis=0)
            print("[faultwhere] getrusage of the caller's thread per step: %.1f minor faults, %.1f us system, %.1f us user, %.2f voluntary / %.2f "
                  "involuntary switches" % tuple(m), file=sys.stderr)
        barrier()
        gc.enable()
        cg1, wall1 = cgroup_cpu(), time.perf_counter()
        host_cpu = None
        if cg0 and cg1:   # CPU time of the whole control group (every thread of the library and of the driver) over the segment's wall time
            host_cpu = {"cpu_seconds_per_second": round((cg1["usage_usec"] - cg0["usage_usec"]) * 1e-6 / max(wall1 - wall0, 1e-9), 2),
                        "cfs_periods_throttled": cg1["nr_throttled"] - cg0["nr_throttled"], "cfs_periods": cg1["nr_periods"] - cg0["nr_periods"]}
        if alt_fit:
            for j, m in enumerate(alt_fit):
                v = per[j::len(alt_fit)]
                print(f"[alternate] fit threads {m}: mean {np.mean(v) * 1e3:.1f} us  p50 {pct(v, 50) * 1e3:.1f}  p99 {pct(v, 99) * 1e3:.1f}  max {np.max(v) * 1e3:.1f}  "
                      f"frames above 1 ms: {int(np.sum(np.asarray(v) > 1.0))}  over {len(v)} frames", file=sys.stderr)
        if alt_spin:
            for j, m in enumerate(alt_spin):
                v = per[j::len(alt_spin)]
                print(f"[alternate] poll {m} us: mean {np.mean(v) * 1e3:.1f} us  p50 {pct(v, 50) * 1e3:.1f}  p99 {pct(v, 99) * 1e3:.1f}  max {np.max(v) * 1e3:.1f}  "
                      f"frames above 1 ms: {int(np.sum(np.asarray(v) > 1.0))}  over {len(v)} frames", file=sys.stderr)
        if alt_img:
            for j, m in enumerate(alt_img):
                v = per[j::len(alt_img)]
                print(f"[alternate] images {m}: mean {np.mean(v) * 1e3:.1f} us  p50 {pct(v, 50) * 1e3:.1f}  p99 {pct(v, 99) * 1e3:.1f}  over {len(v)} frames", file=sys.stderr)
        if alt_knobs:
            pkg.debug_knobs(0)
            for j, m in enumerate(alt_knobs):
                v = per[j::len(alt_knobs)]
                print(f"[alternate] knobs {m}: mean {np.mean(v) * 1e3:.1f} us  p50 {pct(v, 50) * 1e3:.1f}  p99 {pct(v, 99) * 1e3:.1f}  max {np.max(v) * 1e3:.1f}  "
                      f"frames above 1 ms: {int(np.sum(np.asarray(v) > 1.0))}  over {len(v)} frames", file=sys.stderr)
        if alt_modes:
            ctx.update_compression_mode(0)
            for j, m in enumerate(alt_modes):
                v = per[j::len(alt_modes)]
                print(f"[alternate] mode {m}: mean {np.mean(v) * 1e3:.1f} us  p50 {pct(v, 50) * 1e3:.1f}  over {len(v)} frames", file=sys.stderr)
        stats = {k: sm.stats[k] - base.get(k, 0) for k in sm.stats}
        split = {}
        for k, v in sm.tc.total.items():
            if k.startswith("[Time-Cam]"):
                a, c = tc0.get(k, (0.0, 0))
                split[k.replace("[Time-Cam] ", "")] = round((v - a) / max(1, sm.tc.count[k] - c) * 1e3, 4)
        if os.environ.get("PLV_BENCH_FRAMES"):
            print("[frames] ms per step:", " ".join(f"{v:.3f}" for v in per), "| steps that (re)allocated a buffer:", grew, "| host cpu:", host_cpu, file=sys.stderr)
            groups = {}
            for ms_, ch_, sp_, re_, la_, ln_ in per_frame.get("classes", []):
                groups.setdefault(("chained" if ch_ else "unchained", "speculated" if sp_ else ("withdrawn" if re_ else "after-flow"), "lines accepted" if la_ else "no line", int(ln_)), []).append(ms_)
            for k_ in sorted(groups, key=lambda k: -len(groups[k])):
                v_ = groups[k_]
                print(f"[classes] {k_}: {len(v_)} frames, mean {np.mean(v_) * 1e3:.1f} us, p50 {pct(v_, 50) * 1e3:.1f}", file=sys.stderr)
            print("[frames] (ms, msckf features, line pool, lines triangulated, lines accepted, LK iterations, segments detected; us inside: flow wait, point update, line update, worker: maps wait, detection, feed) per step:", per_frame.get("trace"), file=sys.stderr)
            for row in slow:
                print("[slow step] step %d: %.3f ms; inside its parts (ms) %s; plv_camera_frame %.3f, plv_ctx_synchronize %.3f" % row, file=sys.stderr)
        return dict(elapsed=reduce_max(elapsed, dist), per=per, per_frame=per_frame, cnt=cnt, stats=stats, split=split, grew=grew, host_cpu=host_cpu)

    for f in range(PROLOGUE + args.warmup):
        pl.camera(*pl.next_frame())
    if not sm.state.initialized or len(sm.state.clones) < wl["hz"] - 1:
        raise RuntimeError("the filter did not reach a full window during the prologue")
    pl.staged, pl.pinned = args.images == "resident", args.images == "pinned"
    seg = timed_segment(args.steps)                                   # (1) images resident in HBM -> `value`
    pl.staged, pl.pinned = True, False
    if os.environ.get("PLV_BENCH_FINGERPRINT"):     # (determinism checks: the filter's state after the timed segment, to the last bit)
        print("[fingerprint] p = %r  q = %r  trace(P) = %r  routes = %r" % (tuple(float(x) for x in sm.state.imu.p), tuple(float(x) for x in sm.state.imu.q),
                                                                          float(np.trace(ctx.cov_download(sm.state.n))), pkg.route_counts()), file=sys.stderr)
    if os.environ.get("PLV_BENCH_STOP_AFTER_MAIN"):   # (measurement aid: the library's phase table then covers the timed segment only)
        print("[main segment] ms per step %.4f  p50 %.4f" % (seg["elapsed"] / args.steps * 1e3, float(np.median(seg["per"]))), file=sys.stderr)
        sm.close()
        sys.exit(0)
    seg_pcie = seg_pin = seg_res_alt = None
    if not args.no_pcie:
        # (2) host images, measured ALTERNATING with the resident hand-over over the next 3 * steps frames (frame f: resident, any host
        # buffer, the library's pinned block, ...): consecutive segments are other frames of the drive, whose cost differs by more than
        # the hand-over's (round 6: 20-step segments showed the pinned block 40 us SLOWER than any buffer, frame by frame it is 13 us
        # faster) — the three means below come from interleaved frames of one segment
        seg_alt = timed_segment(3 * args.steps, images=("resident", "host", "pinned"))
        pl.staged, pl.pinned = True, False

        def third(j):
            v = [float(x) for x in seg_alt["per"][j::3]]
            return dict(seg_alt, per=v, elapsed=reduce_max(float(np.mean(v)) * 1e-3 * args.steps, dist))
        seg_res_alt, seg_pcie, seg_pin = third(0), third(1), third(2)
    seg_py = timed_segment(npy, through_python=True)                  # (3) as (1), the Python driver's marshalling inside the step
    elapsed, per, per_frame, cnt, stats, split = (seg[k] for k in ("elapsed", "per", "per_frame", "cnt", "stats", "split"))
    n_state = sm.state.n
    spin_us, fit_threads = pkg.line_worker_config()
    variants = None
    if not args.no_variants:
        # non-default library settings over the next frames of the stream (short segments, resident images): (a) the reference's own
        # update route (Householder compression + EKF step); (b) the library's threads blocking at once instead of polling
        # Each variant is measured ALTERNATING with the default, frame by frame, over the next 2 * nvar frames of the stream: later
        # frames are other frames, and a box drifts by +-25 us between segments — more than either variant moves.
        variants = {}

        def alternating(set_variant, set_default):
            v = timed_segment(2 * nvar, hook=lambda f: (set_variant if f % 2 else set_default)())
            set_default()
            return float(np.mean(v["per"][1::2])), float(np.mean(v["per"][0::2])), v
        var_ms, def_ms, v = alternating(lambda: ctx.update_compression_mode(1), lambda: ctx.update_compression_mode(0))
        variants["compression_householder"] = {"ms_per_step": var_ms, "default_ms_per_step_on_the_alternate_frames": def_ms, "frames": nvar,
                                               "what": "plv_update_compression_mode(1): the reference's route (compression of the stacked rows by "
                                                       "Householder reflections, then S = R P R^T + I) instead of the whitened update"}
        # (c) the north star's 21 x 21 LK patch (plv_config.win_size; 15 is the reference's value and the parity setting): every lane of a
        # point's workgroup carries a second window pixel
        var_ms, def_ms, v = alternating(lambda: ctx.set_lk_window(21), lambda: ctx.set_lk_window(15))
        variants["win21"] = {"ms_per_step": var_ms, "default_ms_per_step_on_the_alternate_frames": def_ms, "frames": nvar,
                             "what": "plv_set_lk_window(21): 21 x 21 LK window (441 pixels per point per level) instead of 15 x 15; the tracks differ, "
                                     "so this is a cost figure, not a parity run"}
        var_ms, def_ms, v = alternating(lambda: pkg.line_worker_config(0, -1), lambda: pkg.line_worker_config(spin_us, -1))
        variants["line_threads_blocking"] = {"ms_per_step": var_ms, "default_ms_per_step_on_the_alternate_frames": def_ms, "frames": nvar,
                                            "what": "PLV_LINE_SPIN_US=0: the line worker and the fitters sleep on their condition variables"}

    # ---- roofline leg: HIP events around every kernel launch of the camera step, on the stream each kernel is launched on (a separate
    # pass over the next frames of the stream, same schedule as the timed passes, so that the event records do not perturb them)
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
        ctx.synchronize()
        table = ctx.prof_table()
        s2 = {k: sm.stats[k] - b2.get(k, 0) for k in sm.stats}
        F = s2["cam_features"] / max(1, done)
        L = s2["lines_triangulated"] / max(1, done)
        k_cols = n_state - 15 - 6    # every clone + the intrinsics (the IMU pose of the newest frame excluded)
        M = wl["hz"]
        tracked = int(np.mean(per_frame["tracked"]))
        work = wm.frame_work(wl["w"], wl["h"], ctx.pyramid_levels(0), tracked, cnt["lk_iters"] / args.steps, 15, F, M, k_cols,
                             n_state, L=L, Ml=max(2, M // 3), kl=k_cols, n_new=max(1, wl["num_features"] // M),
                             pool_pts=stats["cam_features"] / max(1, args.steps) * 1.5, pool_lines=stats["line_pool"] / max(1, args.steps))
        kernels = {k: v for k, v in table.items() if v[0] > 0}
        traffic_tab, traffic_src = pmc_table()
        abytes = wm.frame_bytes(wl["w"], wl["h"], ctx.pyramid_levels(0), tracked, cnt["lk_iters"] / args.steps, 15, F, M, k_cols, n_state, L=L,
                                Ml=max(2, M // 3), kl=k_cols, n_new=max(1, wl["num_features"] // M),
                                pool_pts=stats["cam_features"] / max(1, args.steps) * 1.5, pool_lines=stats["line_pool"] / max(1, args.steps),
                                n_clones=wl["hz"] + 1)

        def entry(name):
            n_launch, ms = kernels[name]
            kind, per_launch = work.get(name, ("hbm", 0.0))
            avg_s = ms / max(n_launch, 1) * 1e-3
            if kind == "hbm":
                achieved, peak, unit = per_launch / avg_s / 1e9, wm.HBM_PEAK_GBS, "GB/s"
            else:
                achieved, peak, unit = per_launch / avg_s / 1e12, wm.F64_MFMA_PEAK_TF, "TFLOP/s"
            # (the in-library profiler labels the fused launches by what they carry; rocprofv3 names the __global__ function)
            alias = {"tri_jacobian_nullspace_kernel": "jacobian_nullspace_kernel", "line_tri_jacobian_nullspace_kernel": "line_jacobian_nullspace_kernel",
                     "half_canny_kernel": "canny_kernel"}
            tr = traffic_tab.get(name, traffic_tab.get(alias.get(name, name)))
            ab = abytes.get(name, abytes.get(alias.get(name, name), per_launch if kind == "hbm" else 0.0))
            return {"kernel": name, "bound": kind, "avg_launch_us": round(avg_s * 1e6, 2), "launches_per_frame": round(n_launch / max(1, done), 2),
                    "us_per_frame": round(ms / max(1, done) * 1e3, 2), "algorithmic_per_launch": per_launch, "achieved": achieved, "peak": peak,
                    "unit": unit, "frac": achieved / peak, "traffic": tr,
                    # bytes for EVERY kernel (tools/work_model.py frame_bytes): counter traffic well above them = wasted re-reads
                    "algorithmic_bytes_per_launch": round(ab), "traffic_over_algorithmic": (round(tr / ab, 2) if (tr and ab) else None),
                    "achieved_GBps": round(ab / avg_s / 1e9, 2) if ab else None,
                    "work_model": wm.PROVENANCE.get(name, wm.PROVENANCE.get(alias.get(name, name), "estimate"))}

        # kernels on side streams, next to the chain the frame waits for: the whitened update's prior factor (every update starts one;
        # its result is only consumed when the gate accepts something), the next frame's detection
        side = {"bchol_prior_kernel", "prior_exact_cols_kernel", "prior_gain_kernel", "fast_tiles_kernel", "fast_topk_kernel", "subpix_kernel"}
        order = sorted(kernels, key=lambda k: -kernels[k][1])
        on_path = [k for k in order if k not in side] or order
        top = entry(on_path[0])
        roof = {"bound": top["bound"], "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
                "traffic": top["traffic"], "traffic_source": traffic_src, "kernel": top["kernel"], "avg_launch_us": top["avg_launch_us"],
                "algorithmic_per_launch": top["algorithmic_per_launch"], "algorithmic_bytes_per_launch": top["algorithmic_bytes_per_launch"],
                "traffic_over_algorithmic": top["traffic_over_algorithmic"], "work_model": top["work_model"],
                "launches_per_frame": round(sum(v[0] for v in kernels.values()) / max(1, done), 1),
                "kernel_us_per_frame_total": round(sum(v[1] for v in kernels.values()) / max(1, done) * 1e3, 1),
                "per_kernel": [dict(entry(k), on_the_critical_stream=k not in side) for k in order[:10]],
                "kernels_us_per_frame": {k: round(kernels[k][1] / max(1, done) * 1e3, 2) for k in order},
                "note": "the headline entry is the kernel with the most time per frame on the stream the frame waits for; per_kernel lists the "
                        "top ten of all streams (on_the_critical_stream = false: side-stream work that overlaps it).  traffic = counter bytes per dispatch from the committed PMC pass named in traffic_source (an earlier run of this "
                        "workload: same kernels, sizes of that run); mfma-class entries are latency-bound fp64 chains, their frac is against the "
                        "dense fp64 MFMA peak"}
    sm.close()

    stress = None
    if rank == 0 and not args.no_stress:
        stress = stress_update(pkg, device, max(20, min(100, args.steps)), not args.no_cpu)

    cpu = None
    if rank == 0 and not args.no_cpu:
        if cpus_before is not None:
            os.sched_setaffinity(0, cpus_before)   # (the CPU baseline gets every core the process is allowed, not the GPU path's L3 complex)
        cpu = cpu_baseline(wl, stream, args.cpu_frames, args.cpu_budget_s, [1, 4, max(1, min(16, os.cpu_count() or 1))])

    if rank == 0:
        mean = lambda a: (round(float(np.mean(a)), 1) if len(a) else None)
        what = (f"{wl['w']}x{wl['h']} mono, {mean(per_frame['tracked'])} KLT points tracked per frame (TrackKLT n_pts {wl['num_features']}; 15x15 "
                "window, 5 pyramid levels)")
        if wl["lines"]:
            what += (f" + line front-end (half-resolution Canny + fast line detector: {cnt['lines_detected'] / args.steps:.1f} segments detected, "
                     f"{mean(per_frame['kept'])} kept per frame by the reference's point-line assignment (TrackLSD.cpp:744-792, bounding-box "
                     "test with its end-point mix-up kept as is)")
        ms_step = elapsed / args.steps * 1e3
        vs = {}
        if cpu is not None:
            one = cpu["detail"]["1_thread"]
            inside_ms = (cnt["frame_ns"] + cnt["sync_ns"]) / args.steps * 1e-6
            vs = {"cpu_compiled_ms_per_frame": one["inside_mean_ms"], "speedup_resident": one["inside_mean_ms"] / ms_step,
                  "hip_ms_per_frame_inside_the_library": inside_ms, "speedup_inside_the_libraries": one["inside_mean_ms"] / inside_ms,
                  "note": "speedup_resident = CPU frame timed inside liboracle.so / HIP step timed by this Python driver around its call (what "
                          "`value` is); speedup_inside_the_libraries compares like with like: steady_clock inside plv_camera_frame + "
                          "plv_ctx_synchronize against steady_clock inside orc_frame_camera_frame"}
            if seg_pcie is not None:
                vs["speedup_pcie_inclusive"] = one["inside_mean_ms"] / (seg_pcie["elapsed"] / args.steps * 1e3)
            for key, d in cpu["detail"].items():
                if key != "1_thread":
                    vs[f"speedup_vs_cpu_{key}"] = d["inside_mean_ms"] / ms_step
            vs["speedup_vs_cpu_all_cores"] = cpu["ms_per_frame_all_cores"] / ms_step
            vs["cpu_all_cores_threads"] = cpu["cores_all_cores"]
        line = {
            "metric": ("frames/sec (track+EKF update), 752x480 mono, 250 pts+80 lines; ATE vs CPU ref" if args.workload in ("C", "C_avenue") else
                       f"frames/sec (track+EKF update), {wl['w']}x{wl['h']} mono, {wl['points']} pts" + (" + lines" if wl["lines"] else "")),
            "value": args.steps * world / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            # scalars first (whatever keeps only the scalar fields of this line still gets them); `value` is the
            # resident-image step (the contract), *_pcie_inclusive the drop-in adapter's call (host cv::Mat pointer, slot -1)
            "latency_p50_ms": pct(per, 50), "latency_p99_ms": pct(per, 99),
            # what the step occupies on the host: CPU seconds per second of wall time over the first timed segment (the
            # whole control group: caller, line worker, helper threads — they poll), the threads, and where they were pinned
            "host_cpu_seconds_per_second": None if not seg.get("host_cpu") else seg["host_cpu"]["cpu_seconds_per_second"],
            "device_bytes_peak": pkg.memory_bytes()["device_peak"], "pinned_bytes_peak": pkg.memory_bytes()["pinned_peak"],
            "host_threads": (2 + fit_threads) if wl["lines"] else 1,
            "host_threads_what": (f"1 caller + 1 line worker + {fit_threads} polling helper threads of the line detector's host stage" if wl["lines"] else "1 caller"),
            "cpu_affinity_mode": pinned.get("mode"), "cpu_affinity_cpus": pinned.get("cpus"),
            "cpu_baseline_affinity": "every CPU the process was started with (the GPU path's pin is lifted before the CPU baseline runs)",
            "methodology": "round 5 on: timed region = plv_camera_frame + plv_ctx_synchronize on pre-marshalled arguments (ms_per_step_python = with the Python "
                           "driver's marshalling inside, what rounds 1-4 reported); process + library threads pinned to one L3 complex next to the GPU (--pin); "
                           "60 untimed prologue frames; round 6: value_pcie_inclusive (any host buffer) and value_pinned_host_image (plv_image_buffer) beside the "
                           "resident-image `value`",
            "tracked_points_per_frame": mean(per_frame["tracked"]), "lines_kept_per_frame": mean(per_frame["kept"]) if wl["lines"] else 0,
            "lines_accepted_per_frame": round(stats["lines_accepted"] / args.steps, 2), "msckf_features_per_frame": round(stats["cam_features"] / args.steps, 2),
            "ms_per_step_python": seg_py["elapsed"] / npy * 1e3,
            "ms_per_step_pcie_inclusive": None if seg_pcie is None else seg_pcie["elapsed"] / args.steps * 1e3,
            "value_pcie_inclusive": None if seg_pcie is None else args.steps * world / seg_pcie["elapsed"],
            # the image in the library's page-locked block when the step starts (plv_image_buffer: the adapters' route): the transfer is inside
            "ms_per_step_pinned_host_image": None if seg_pin is None else seg_pin["elapsed"] / args.steps * 1e3,
            "ms_per_step_resident_on_the_alternate_frames": None if seg_res_alt is None else seg_res_alt["elapsed"] / args.steps * 1e3,
            "value_pinned_host_image": None if seg_pin is None else args.steps * world / seg_pin["elapsed"],
            "vs_cpu_all_cores_pinned_host_image": (None if (cpu is None or seg_pin is None) else cpu["ms_per_frame_all_cores"] / (seg_pin["elapsed"] / args.steps * 1e3)),
            "vs_cpu_1_thread": vs.get("speedup_resident"), "vs_cpu_all_cores": vs.get("speedup_vs_cpu_all_cores"),
            "vs_cpu_all_cores_pcie_inclusive": (None if (cpu is None or seg_pcie is None) else cpu["ms_per_frame_all_cores"] / (seg_pcie["elapsed"] / args.steps * 1e3)),
            "cpu_all_cores_threads": None if cpu is None else cpu["cores_all_cores"],
            "config": {
                "workload": f"BASELINE {wl['cfg']}: {what}; {wl['hz']}-clone window ({wl['hz']} Hz camera and clones, 1 s), n = {n_state}; "
                            f"rendered drive ('{wl['scene']}' scene, camera mount pitch {wl['mount'][0]:g} / yaw {wl['mount'][1]:g} deg) with IMU + wheel odometry; the update consumes the tracker's own database",
                "step": "plv_camera_frame = plv_tracker_feed[_staged] -> plv_vanishing_points + plv_line_tracker_feed -> plv_camera_try_update "
                        "(plv_camera_update_points -> plv_camera_get_line_features -> dx applied -> plv_camera_update_lines -> dx applied), then "
                        "plv_ctx_synchronize (ctx stream, detection side stream, line worker); sequential; IMU propagation, cloning, "
                        "marginalisation and wheel updates run between the steps, untimed",
                "replicas": world, "n_state": n_state,
                # (scalar copies of what the nested blocks below hold)
                "tracked_points_per_frame": mean(per_frame["tracked"]), "lines_kept_per_frame": mean(per_frame["kept"]) if wl["lines"] else 0,
                "msckf_features_per_update": round(stats["cam_features"] / args.steps, 2), "lines_triangulated_per_frame": round(stats["lines_triangulated"] / args.steps, 2),
                "kernel_launches_per_frame": round(cnt["launches"] / args.steps, 1), "host_synchronisations_per_frame": round(cnt["syncs"] / args.steps, 1),
                "line_launches_chained_per_frame": round(cnt.get("chained", 0) / args.steps, 2),
                "ms_per_step_inside_the_library": round((cnt["frame_ns"] + cnt["sync_ns"]) / args.steps * 1e-6, 4),
                "us_per_step_in_plv_ctx_synchronize": round(cnt["sync_ns"] / args.steps * 1e-3, 1),
                "cpu_affinity": pinned,
                "timed_region": "plv_camera_frame + plv_ctx_synchronize on arguments marshalled beforehand (SystemManager.camera_prepare): two ctypes calls; "
                                "ms_per_step_python = the same step with the Python driver's marshalling inside it, over the next frames",
                "ms_per_step_python": round(seg_py["elapsed"] / npy * 1e3, 4),
                "ms_per_step_pcie_inclusive": None if seg_pcie is None else round(seg_pcie["elapsed"] / args.steps * 1e3, 4),
                "ms_per_step_pinned_host_image": None if seg_pin is None else round(seg_pin["elapsed"] / args.steps * 1e3, 4),
                "host_threads": {"caller": 1, "library_line_worker": 1 if wl["lines"] else 0,
                                 "library_segment_fitters": (min(fit_threads, 2 if (wl["w"] // 2) * (wl["h"] // 2) >= 60000 else 1) if wl["lines"] else 0),
                                 "library_segment_fitters_configured_maximum": fit_threads if wl["lines"] else 0, "poll_before_blocking_us": spin_us,
                                 "note": "the library's threads run the line detector's host stage (chain walk + segment growth) and the line "
                                         "tracker's bookkeeping next to the caller's thread; a waiting thread polls for poll_before_blocking_us, "
                                         "then blocks (plv_line_worker_config); cpu_baseline.detail has the CPU frame at 1, 4 and 16 threads"},
                "compression": {"mode": "whitened update (plv_update_compression_mode 0): information matrix of the accepted rows + factor "
                                        "of the prior block on a side stream; no factor of the measurements",
                                "frames_whose_last_update_took_it": cnt["whitened_frames"], "frames": args.steps,
                                "updates_by_route": {k: v for k, v in zip(("uncompressed", None, "householder", None, "whitened",
                                                                          "whitened_rejected_then_householder"), cnt["routes"][:6]) if k},
                                "note": "agrees with the Givens oracle to 1e-10 (P') and 1e-9 (dx) on every captured replay batch and up to "
                                        "condition 1e8 (tests/test_gpu_update_hard.py); config.variants.compression_householder is the "
                                        "reference's route (mode 1)"},
                "variants": variants,
                "host_cpu": seg.get("host_cpu"),
                "latency_ms": {"mean": float(np.mean(per)), "p50": pct(per, 50), "p99": pct(per, 99), "max": float(np.max(per))},
                "pcie_inclusive": None if seg_pcie is None else {
                    "what": "second timed segment, the next 3 x steps frames of the stream, the image's hand-over ALTERNATING frame by frame: resident "
                            "(resident_on_the_alternate_frames), any host buffer (this entry: plv_tracker_feed, its "
                            f"{wl['w'] * wl['h'] // 1024} KB host copy into the library's page-locked block and the PCIe transfer inside the step), "
                            "the image already in that block (pinned_host_image: plv_image_buffer, the transfer inside the step)",
                    "resident_on_the_alternate_frames": None if seg_res_alt is None else {"ms_per_step": seg_res_alt["elapsed"] / args.steps * 1e3,
                        "latency_ms": {"mean": float(np.mean(seg_res_alt["per"])), "p50": pct(seg_res_alt["per"], 50), "p99": pct(seg_res_alt["per"], 99)}},
                    "pinned_host_image": None if seg_pin is None else {"ms_per_step": seg_pin["elapsed"] / args.steps * 1e3,
                        "latency_ms": {"mean": float(np.mean(seg_pin["per"])), "p50": pct(seg_pin["per"], 50), "p99": pct(seg_pin["per"], 99)}},
                    "value": args.steps * world / seg_pcie["elapsed"], "ms_per_step": seg_pcie["elapsed"] / args.steps * 1e3,
                    "latency_ms": {"mean": float(np.mean(seg_pcie["per"])), "p50": pct(seg_pcie["per"], 50), "p99": pct(seg_pcie["per"], 99)}},
                "vs_cpu": vs,
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
                                          "lk_iterations": round(cnt["lk_iters"] / args.steps),
                                          "line_launches_chained_behind_the_point_update": round(cnt.get("chained", 0) / args.steps, 2)},
                "updates": {"point_updates": stats["cam_updates"], "line_updates": stats["line_updates"], "not_psd": stats["not_psd"],
                            "frames": args.steps},
                "images": "resident in HBM (plv_image_stage between the steps); config.pcie_inclusive has the host-buffer variant",
                "front_end_arithmetic": "u8/int16/int64 exact + f32 2x2 solve", "update_arithmetic": "f64",
                "stress": stress,
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
