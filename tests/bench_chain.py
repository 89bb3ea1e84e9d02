#!/usr/bin/env python3
"""tests/bench_chain.py (round 1's bench.py) — kernel-chain rate: frames/sec of the per-frame (track + EKF update) hot path on MI355X.

One "step" = one camera frame of the configuration BASELINE.json's metric is quoted on ("752x480 mono, 250 pts+80 lines" =
configs[2]: configs[1]'s point path plus the line front-end and the line update), every input already resident in HBM when
the timed region starts (`--workload B` times configs[1], points only; its rate is also reported in `config.points_only`):
    plv_feed_staged          equalizeHist + 5-level pyramid of the staged 752x480 image
    plv_perform_matching     15x15 pyramidal LK on 250 points, radtan undistort, 7-point RANSAC
    plv_build_jacobians_resident  70 features x 15 observations: FEJ clone-polynomial interpolation,
                             projection / distortion Jacobians, whitening (inputs uploaded per frame)
    plv_cov_rollback         (restores P so that every step does identical work)
    plv_msckf_update_resident  Givens nullspace, chi2 gate, compression, EKFUpdate on the n = 113
                             covariance (fp64)
    plv_line_tracker_feed_points  TrackLSD::feed_monocular: half-resolution Canny + fast line detector, point-line assignment
                             with the frame's tracked points, line matching, undistortion, classification, track store
    plv_build_line_jacobians_resident + plv_msckf_update_resident  80 lines x 15 observations (Pluecker Jacobians 30 x 6,
                             null space of 6, chi2-only gate), second EKFUpdate of the frame as UpdaterCamera::try_update
Feature selection and triangulation (host logic in the reference) are outside the timed step.

Contract: `python bench.py --gpus N --steps K --warmup W`; N > 1 is launched by torchrun, one rank
per GPU.  The path does not shard ("replicas only"), so N ranks run N independent
replicas and `value` is their aggregate frames/s ("weak" scaling); no data-path collective.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F64_MFMA_PEAK_TF = 78.6    # vendor FP64 matrix spec; v_mfma_f64_16x16x4_f64
W, H = 752, 480
N_PTS, WIN = 250, 15
N_STATE, K_COLS, F_FEATS, M_OBS, FDIM = 113, 98, 70, 15, 3
N_LINES, LINE_LD, LINE_EDGES = 80, 32, 200
SIGMA2 = 2.25  # the gate's R = sigma_pix^2 I on rows that are already whitened (UpdaterCamera.cpp:237-238), kept as is


def build_inputs(with_lines=False):
    import synth
    # SURVEY §8(d) cfg 3: the cfg 2 stream + straight high-contrast edges rendered before the blur
    canvas = synth.texture_canvas(W, H, seed=42, lines=LINE_EDGES if with_lines else 0)
    frames = [synth.render_frame(canvas, W, H),
              synth.render_frame(canvas, W, H, tx=4.2, ty=-3.1, rot_deg=0.3, scale=1.002)]
    pts = synth.grid_points(W, H, N_PTS, seed=5, border=16)
    P = synth.spd_cov(N_STATE)
    # filter problem of SURVEY §8(d) cfg 2: 15 clones on a 1 m/s arc, 70 landmarks x 15 observations,
    # intrinsics calibrated online -> n = 113, k = 98.  0.4 px noise so that the reference's norm gate
    # (whitened |res| < 3, UpdaterCamera.cpp:242) lets most of the 15-observation tracks through.
    scene = synth.vio_scene(n_clones=15, F=F_FEATS, M=M_OBS, seed=3, noise_px=0.4)
    assert scene["n_state"] == N_STATE
    return frames, pts, P, scene


def points_on_lines(lines, base_pts, n_lines):
    """250 point positions of which up to two lie on each detected segment (FAST corners sit on such edges and their ends; TrackLSD
    keeps only lines that own a point, TrackLSD.cpp:744-792), the rest from the grid set.  The positions honour the reference's
    bounding-box test, which reads (x1, y1, x2, y2) as (lx1, lx2, ly1, ly2) (TrackLSD.cpp:753-764): a line whose own points all
    fail that test cannot be kept by any tracker and gets none."""
    length = np.hypot(lines[:, 2] - lines[:, 0], lines[:, 3] - lines[:, 1])
    on = []
    for i in np.argsort(-length, kind="stable"):
        x1, y1, x2, y2 = lines[i]
        inside = [sgm for sgm in np.linspace(0.04, 0.96, 47)
                  if min(x1, y1) + 3 <= x1 + sgm * (x2 - x1) <= max(x1, y1) - 3 and min(x2, y2) + 3 <= y1 + sgm * (y2 - y1) <= max(x2, y2) - 3
                  and 16 < x1 + sgm * (x2 - x1) < W - 16 and 16 < y1 + sgm * (y2 - y1) < H - 16]
        for sgm in ([inside[len(inside) // 4], inside[(3 * len(inside)) // 4]] if len(inside) >= 2 else inside):
            on.append((x1 + sgm * (x2 - x1), y1 + sgm * (y2 - y1)))
        if len(on) >= 2 * n_lines:
            break
    on = np.array(on, dtype=np.float32).reshape(-1, 2)
    return np.concatenate([on, base_pts[:N_PTS - len(on)]]).astype(np.float32), len(on)


def update_work(F, rows_f, fdim, k, n, qr_launches=1):
    """Algorithmic flops per launch of the update kernels for F features of rows_f rows (before the null-space projection removes
    fdim of them) on k columns of an n-state filter (formulas of SURVEY.md §8(d), restated in DESIGN.md)."""
    mp = rows_f - fdim
    m = F * mp
    nc = k + 1
    r = k
    return {
        "nullspace_kernel": F * 6.0 * (fdim + k + 1) * (rows_f * fdim - fdim * (fdim + 1) / 2),
        "chi2_gate_kernel": F * (2.0 * mp * mp * k + mp ** 3 / 3.0),
        "chi2_t_kernel": F * 2.0 * mp * k * k,
        "qr_accum_kernel": (2.0 * m * nc * nc - (2.0 / 3) * nc ** 3) / max(1, qr_launches),
        "gram_chunk_kernel": 1.0 * m * nc * nc,  # upper tiles only: half of 2 m nc^2
        "gram_reduce_kernel": (m / 64.0) * (nc * nc / 2.0) * 8,
        "bchol_compress_kernel": nc ** 3 / 3.0,
        "bchol_ekf_kernel": r ** 3 / 3.0 + 1.0 * r * r * (n + 1),
        "gather_cov_kernel": 2.0 * (k * n + k * k) * 8,
        "ekf_dc_kernel": 1.0 * n * n * r + 2.0 * n * r,
        "ekf_commit_kernel": 3.0 * n * n * 8,
        "ekf_mt_kernel": 2.0 * n * k * r,
        "ekf_s_kernel": 2.0 * r * r * k,
    }


def algorithmic_work(levels, lk_iters_per_frame, qr_launches, with_lines=False, k_lines=90):
    """Per-LAUNCH algorithmic bytes (HBM-class kernels) or flops (dense fp64 kernels); formulas
    from SURVEY.md §8(d), restated in DESIGN.md.  With lines, the kernels both updates of a frame launch carry the mean of the
    two launches (70 x 27 rows on 98 columns, 80 x 24 rows on 90 columns)."""
    lv = levels
    it_per_pl = lk_iters_per_frame / float(N_PTS * lv)
    pyr_reads_writes = (4.0 / 3 + 1.0 / 3) * W * H
    hbm = {"gram_reduce_kernel", "gather_cov_kernel", "ekf_commit_kernel"}
    up = update_work(F_FEATS, 2 * M_OBS, FDIM, K_COLS, N_STATE, qr_launches)
    if with_lines:
        ul = update_work(N_LINES, 2 * M_OBS, 6, k_lines, N_STATE, qr_launches)
        up = {kname: (0.5 * (v + ul[kname]) if kname != "nullspace_kernel" else ul[kname]) for kname, v in up.items()}
    work = {
        "hist_kernel": ("hbm", W * H),
        "equalize_kernel": ("hbm", 2 * W * H),
        "pyrdown_kernel": ("hbm", pyr_reads_writes / max(1, lv - 1)),
        "pyrdown2_kernel": ("hbm", pyr_reads_writes / 2.0),
        "lk_kernel": ("hbm", N_PTS * lv * ((WIN + 2) ** 2 + it_per_pl * (WIN + 1) ** 2) + N_PTS * 17),
        "undistort_kernel": ("hbm", 2 * N_PTS * 16),
        "half_kernel": ("hbm", W * H + W * H / 4.0),
        "canny_kernel": ("hbm", 2 * W * H / 4.0),
        "jacobian_kernel": ("mfma", F_FEATS * M_OBS * 3000.0),
        "jacobian_nullspace_kernel": ("mfma", F_FEATS * M_OBS * 3000.0 + update_work(F_FEATS, 2 * M_OBS, FDIM, K_COLS, N_STATE)["nullspace_kernel"]),
        "line_jacobian_kernel": ("mfma", N_LINES * M_OBS * 6000.0),
        "ransac_hyp_kernel": ("mfma", 1000 * 3 * N_PTS * 40.0),
        "ransac_select_kernel": ("mfma", N_PTS * 40.0),
    }
    for kname, v in up.items():
        work[kname] = ("hbm" if kname in hbm else "mfma", v)
    return work


def cpu_baseline(pkg, frames, pts_of, P, scene, sample_frames, lt=None, cols_l=None, vps=None):
    """The CPU oracle (fp64 / OpenCV-contract restatement, g++ -O3, 1 thread) timed on this host on a
    bounded sample of the same workload.  kind = "port": the upstream binary cannot be built
    (Eigen/OpenCV/Boost/ROS absent — DESIGN.md)."""
    import oracle_lib
    import synth
    orc, fo, jo = oracle_lib.load(), oracle_lib.load_front(), oracle_lib.load_jac(pkg)
    lo = oracle_lib.load_line() if lt is not None else None
    st, tr = synth.scene_views(pkg, scene)
    cols = jo.columns(st, tr)
    q95 = synth.q95_table()
    K8 = synth.EUROC_K8
    ids = np.arange(1, N_PTS + 1, dtype=np.uint64)
    eq_prev = fo.equalize_hist(frames[0])
    prev = fo.pyramid(eq_prev)
    last = None
    t_front = t_upd = t_lfront = t_lupd = 0.0
    for i in range(sample_frames):
        p0 = pts_of[i & 1]
        t0 = time.perf_counter()
        eq = fo.equalize_hist(frames[(i + 1) & 1])
        cur = fo.pyramid(eq)
        out = fo.perform_matching(prev, cur, p0, p0, K8, nthreads=1)
        t1 = time.perf_counter()
        if lo is not None:   # TrackLSD::feed_monocular: detection, assignment, matching, undistortion, classification
            lines = lo.detect_lines(eq)
            a = lo.assign_points_to_lines(lines, out[1], ids)
            kept = lines[a["kept"]]
            if last is not None and len(kept):
                lo.line_match(kept, a["rel_ptr"], a["rel_id"], last[0], last[1], last[2])
                fo.undistort(K8, kept.reshape(-1, 2))
                for q in range(len(kept)):
                    lo.line_classification(kept[q], vps)
            last = (kept, a["rel_ptr"], a["rel_id"])
        t2 = time.perf_counter()
        rows, Hf, Hx, res = jo.build_jacobians(st, tr, cols, 2 * M_OBS)
        rc, P1, _, _, _ = orc.msckf_update(P, rows, Hf, Hx, res, cols, SIGMA2, q95)
        t3 = time.perf_counter()
        if lo is not None:
            rows, Hf, Hx, res = jo.build_line_jacobians(st, lt, cols_l, LINE_LD)
            orc.msckf_update(P1, rows, Hf, Hx, res, cols_l, SIGMA2, q95, res_norm_gate=0.0)
        t4 = time.perf_counter()
        prev = cur
        t_front += t1 - t0
        t_lfront += t2 - t1
        t_upd += t3 - t2
        t_lupd += t4 - t3
    tot = t_front + t_upd + t_lfront + t_lupd
    ms = lambda x: x / sample_frames * 1e3
    # the one stage the reference runs through cv::parallel_for_ (the per-point LK), with the host's cores
    nthr = max(1, min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    for i in range(4):
        fo.perform_matching(prev, cur, pts_of[i & 1], pts_of[i & 1], K8, nthreads=nthr)
    t_mt = (time.perf_counter() - t0) / 4
    parts = f"point front-end {ms(t_front):.2f} ms + point update {ms(t_upd):.2f} ms"
    if lo is not None:
        parts += f" + line front-end {ms(t_lfront):.2f} ms + line update {ms(t_lupd):.2f} ms"
    return {"value": sample_frames / tot, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{sample_frames} frames of the same workload ({parts} per frame), oracle g++ -O3 single thread, "
                      f"host has {os.cpu_count()} cores; LK + RANSAC alone with {nthr} threads: {t_mt * 1e3:.2f} ms"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate runs, profiles/r01/README.md); None when the summary is missing.  The guide's gfx950
    correction (FETCH_SIZE x 2) applies to wide coalesced reads only; byte / 8-byte gathers as in lk_kernel are
    uncalibrated, so the raw counter is reported."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_*_pmc_hbm.csv")))
    if not found:
        return None, None
    path = found[-1]  # the most recent round / letter
    rel = os.path.relpath(path, ROOT)
    with open(path) as fh:
        next(fh)
        for line in fh:
            k, n, f_kb, w_kb = line.strip().split(",")
            if k.split("<")[0].endswith(kernel):
                return (float(f_kb) + float(w_kb)) * 1024.0, rel + " (FETCH_SIZE + WRITE_SIZE, KB)"
    return None, None


def reduce_max(elapsed, dist):
    """MAX over ranks of the timed region (the replicas exchange nothing else)."""
    if dist is None:
        return elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--cpu-frames", type=int, default=12)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercises the multi-process plumbing only (tests/test_bench_dist.py)")
    ap.add_argument("--update-graph", type=int, default=0,
                    help="1: replay the update launch sequence as a hipGraph (plv_update_graph_mode)")
    ap.add_argument("--workload", choices=["B", "C"], default="C",
                    help="C (default) = BASELINE configs[2], 250 points + 80 lines: the configuration the metric is quoted on; "
                         "B = configs[1], points only")
    ap.add_argument("--sequential", action="store_true",
                    help="one context, update of frame i finished before the front-end of frame i+1 starts")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # torch.distributed is plumbing only (barrier + max-reduce of the wall time); the replicas
        # exchange no data.  gloo on CPU tensors: the HIP work is entirely inside libplviwo_hip.so.
        import torch
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist = dist_mod

    if args.dry_run:
        # replicas only: each rank "processes" its own frames (here: sleeps a rank-dependent time), then the
        # barrier / max-over-ranks / rank-0 JSON logic below it is the same code the real run uses
        barrier_fn = (lambda: dist.barrier()) if dist is not None else (lambda: None)
        barrier_fn()
        t0 = time.perf_counter()
        time.sleep(0.05 * (1 + rank))
        elapsed = reduce_max(time.perf_counter() - t0, dist)
        barrier_fn()
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "value": args.steps * world / elapsed, "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "elapsed_s": elapsed,
                              "device_of_rank": local_rank if world > 1 else 0}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    cfg = pkg.default_config(W, H)
    ndev = max(1, pkg.load_library().plv_device_count())
    cfg.device = (local_rank % ndev) if world > 1 else 0   # one GPU per rank; wraps only when a node has fewer GPUs than ranks
    # Two contexts = two HIP streams, as the reference has two objects (TrackKLT / the updater): `ctx` tracks,
    # `uctx` owns the covariance and runs the update.  In the default (pipelined) mode the update of frame i
    # is enqueued first and the front-end of frame i+1 runs while it executes (tracking does not depend on
    # the filter state, TrackKLT.cpp:96-200), so a step still does one full front-end and one full update.
    ctx = pkg.Context(cfg)
    uctx = ctx if args.sequential else pkg.Context(cfg)

    with_lines = args.workload == "C"
    frames, pts, P, scene = build_inputs(with_lines)
    import synth
    st, tr = synth.scene_views(pkg, scene)
    cols = ctx.jacobian_columns(st, tr)
    assert len(cols) == K_COLS
    ctx.image_stage(0, frames[0])
    ctx.image_stage(1, frames[1])
    uctx.cov_upload(P)
    uctx.cov_checkpoint()
    if args.update_graph:
        uctx.update_graph_mode(1)
    ctx.feed_staged(0)

    state = {}
    pts_of = [pts, pts]                 # positions in frame 0 / frame 1 (the "last" image of even / odd steps)
    ids = np.arange(1, N_PTS + 1, dtype=np.uint64)
    lt = cols_l = vps = None
    if with_lines:
        # untimed set-up: the 80 longest segments of each frame get two tracked points each, so that ~80 lines survive the
        # point-line assignment every frame (SURVEY §8(d) cfg 3: "+80 lines kept after assignment")
        ctx.feed_staged(1)
        lines1 = ctx.detect_lines(0)    # PLV_PYR_CUR = frame 1
        lines0 = ctx.detect_lines(1)    # PLV_PYR_LAST = frame 0
        # positions ON the lines of the image a step tracks INTO, carried back into the image it tracks FROM through the known
        # frame-to-frame warp: pts_of[f] are positions in frame f whose tracked positions lie on the other frame's lines
        q0, q1 = points_on_lines(lines0, pts, N_LINES)[0], points_on_lines(lines1, pts, N_LINES)[0]
        th, c0, tr_, sc_ = np.deg2rad(0.3), np.array([W / 2.0, H / 2.0]), np.array([4.2, -3.1]), 1.002
        Rw = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        fwd = lambda p: ((p - c0) @ Rw.T) * sc_ + c0 + tr_          # frame 0 -> frame 1 (synth.warp_points)
        inv = lambda q: ((q - c0 - tr_) / sc_) @ Rw + c0
        clip = lambda p: np.clip(p, 16, [W - 17, H - 17]).astype(np.float32)
        pts_of = [clip(inv(q1)), clip(fwd(q0))]
        ctx.feed_staged(0)
        ls = synth.line_scene(scene, L=N_LINES, M=M_OBS, noise_px=0.4)
        lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=ls["lines"])
        cols_l = ctx.line_jacobian_columns(st, lt)
        vps = ctx.vanishing_points(scene["R_ItoC"], scene["K8"])

    def line_front_end(c, i, pts_cur):
        """TrackLSD::feed_monocular for the current image with the frame's tracked points; the track store is emptied once per
        window as LineHelper::cleanup_lines prunes it in the reference (LineHelper.cpp:522-553)."""
        c.line_tracker_feed_points(float(i), vps, pts_cur, ids)
        if i % M_OBS == M_OBS - 1:
            c.line_db_remove(c.line_db_ids())

    def line_update(c):
        c.build_line_jacobians_resident(st, lt, cols_l, LINE_LD)
        rc, dx, acc, nr = c.msckf_update_resident(N_STATE, SIGMA2, res_norm_gate=0.0)
        if rc != 0:
            raise RuntimeError("line EKF update rejected inside the benchmark")
        state["accepted_lines"] = int(acc.sum())

    def step_sequential(i, c=None, lines=None):
        c = c or ctx
        lines = with_lines if lines is None else lines
        p0 = pts_of[i & 1] if lines else pts
        c.feed_staged((i + 1) & 1)
        if lines:
            c.line_detect_launch(0)
            c.perform_matching_launch(p0, p0)
            c.line_detect_finish(0)
            out = c.perform_matching_wait()
            line_front_end(c, i, out[0])
        else:
            out = c.perform_matching(p0, p0)
        c.cov_rollback()   # (before the Jacobians: their launch also gathers the covariance blocks of the update)
        c.build_jacobians_resident(st, tr, cols, 2 * M_OBS)
        rc, dx, acc, nr = c.msckf_update_resident(N_STATE, SIGMA2)
        if rc != 0:
            raise RuntimeError("EKF update rejected inside the benchmark")
        if lines:
            line_update(c)
        state["tracked"] = int(out[1].sum())
        state["lk_iters"] = out[4]
        state["accepted"] = int(acc.sum())

    def step_pipelined(i):
        # (enqueueing the front-end between the Jacobians and the rest of the update — plv_perform_matching_launch /
        # _wait — was measured 5-9 % slower than this order on the same box: the update chain is the long one and
        # every launch in front of it delays it)
        uctx.cov_rollback()                                       # update of frame i: enqueue only
        uctx.build_jacobians_resident(st, tr, cols, 2 * M_OBS)
        uctx.msckf_update_resident_launch(SIGMA2)
        ctx.feed_staged((i + 1) & 1)                              # front-end of frame i+1 meanwhile
        out = ctx.perform_matching(pts, pts)
        rc, dx, acc, nr = uctx.msckf_update_resident_wait(N_STATE)
        if rc != 0:
            raise RuntimeError("EKF update rejected inside the benchmark")
        state["tracked"] = int(out[1].sum())
        state["lk_iters"] = out[4]
        state["accepted"] = int(acc.sum())

    def step_pipelined_lines(i):
        # points: as step_pipelined.  Lines: the line update of frame i is enqueued once the point update has been collected (both
        # work on the one covariance, in this order, UpdaterCamera.cpp:139-195) and runs while the line front-end of frame i+1 does.
        p0 = pts_of[i & 1]
        uctx.cov_rollback()
        uctx.build_jacobians_resident(st, tr, cols, 2 * M_OBS)
        uctx.msckf_update_resident_launch(SIGMA2)
        ctx.feed_staged((i + 1) & 1)
        ctx.line_detect_launch(0)             # resize + Canny + copies of the new image, then LK behind them on the stream:
        ctx.perform_matching_launch(p0, p0)   # the host walks the edge chains while the device tracks the points
        ctx.line_detect_finish(0)
        out = ctx.perform_matching_wait()
        rc, dx, acc, nr = uctx.msckf_update_resident_wait(N_STATE)
        if rc != 0:
            raise RuntimeError("EKF update rejected inside the benchmark")
        uctx.build_line_jacobians_resident(st, lt, cols_l, LINE_LD)
        uctx.msckf_update_resident_launch(SIGMA2, res_norm_gate=0.0)
        line_front_end(ctx, i, out[0])
        rc, dx, acc_l, nr = uctx.msckf_update_resident_wait(N_STATE)
        if rc != 0:
            raise RuntimeError("line EKF update rejected inside the benchmark")
        state["tracked"] = int(out[1].sum())
        state["lk_iters"] = out[4]
        state["accepted"] = int(acc.sum())
        state["accepted_lines"] = int(acc_l.sum())

    step = step_sequential if args.sequential else (step_pipelined_lines if with_lines else step_pipelined)

    def barrier():
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    ctx.synchronize()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    ctx.synchronize()
    uctx.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    points_only = None
    if with_lines:
        state["lines_kept"] = len(ctx.line_tracker_last()[1])
        state["lines_detected"] = len(ctx.detect_lines(0))
        if rank == 0 and not args.sequential:   # configs[1] in the same run, for the record (not `value`)
            for i in range(min(args.warmup, 100)):
                step_pipelined(i)
            ctx.synchronize()
            tb = time.perf_counter()
            for i in range(args.steps):
                step_pipelined(i)
            ctx.synchronize()
            uctx.synchronize()
            eb = time.perf_counter() - tb
            points_only = {"workload": "BASELINE configs[1] (250 points, no lines), same run", "value": args.steps / eb, "unit": "frames/s",
                           "ms_per_step": eb / args.steps * 1e3}
    seq_ms = None
    if not args.sequential and rank == 0:  # the un-overlapped frame latency, for the record (not `value`)
        sctx = uctx
        sctx.image_stage(0, frames[0])
        sctx.image_stage(1, frames[1])
        sctx.feed_staged(0)
        nseq = max(20, min(100, args.steps))
        for i in range(10):
            step_sequential(i, sctx)
        sctx.synchronize()
        ts = time.perf_counter()
        for i in range(nseq):
            step_sequential(i, sctx)
        sctx.synchronize()
        seq_ms = (time.perf_counter() - ts) / nseq * 1e3
    elapsed = reduce_max(elapsed, dist)

    # ---- roofline leg: HIP events around every kernel launch on the ctx stream (separate pass so
    # that the event records do not perturb the timed region above)
    roof = None
    if rank == 0:
        nprof = max(20, min(100, args.steps))
        for c in {ctx, uctx}:
            c.prof_enable(True)
            c.prof_reset()
        for i in range(nprof):
            step(i)
        table = {}
        for c in {ctx, uctx}:
            c.prof_enable(False)
            for kname, (cnt, ms) in c.prof_table().items():
                a = table.get(kname, (0, 0.0))
                table[kname] = (a[0] + cnt, a[1] + ms)
        levels = ctx.pyramid_levels(0)
        qr_launches = table.get("qr_accum_kernel", (0, 0))[0] / nprof
        work = algorithmic_work(levels, state["lk_iters"], qr_launches, with_lines, len(cols_l) if with_lines else 90)
        dom = max(table.items(), key=lambda kv: kv[1][1])
        name, (cnt, ms) = dom
        kind, per_launch = work.get(name, ("hbm", 0.0))
        avg_s = ms / max(cnt, 1) * 1e-3
        if kind == "hbm":
            achieved, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = per_launch / avg_s / 1e12, F64_MFMA_PEAK_TF, "TFLOP/s"
        traffic, traffic_src = pmc_traffic(name)
        roof = {"bound": kind, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": name, "avg_launch_us": avg_s * 1e6,
                "algorithmic_per_launch": per_launch,
                "kernels_us_per_frame": {k: round(v[1] / nprof * 1e3, 2) for k, v in
                                         sorted(table.items(), key=lambda kv: -kv[1][1])}}

    cpu = None
    if rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(pkg, frames, pts_of, P, scene, args.cpu_frames, lt, cols_l, vps)

    if rank == 0:
        total_frames = args.steps * world
        line = {
            "metric": "frames/sec (track+EKF update), 752x480 mono, 250 pts+80 lines; ATE vs CPU ref" if with_lines else
                      "frames/sec (track+EKF update), 752x480 mono, 250 pts; ATE vs CPU ref",
            "value": total_frames / elapsed,
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
            "config": {"workload": ("BASELINE configs[2]: 752x480 mono, 250 KLT points (15x15 window, 5 pyramid levels) + line front-end "
                                    "(half-resolution Canny + fast line detector, ~80 lines kept after the point-line assignment), MSCKF "
                                    "update of 70 features x 15 clones then of 80 lines x 15 clones on n=113" if with_lines else
                                    "BASELINE configs[1]: 752x480 mono, 250 KLT points (15x15 window, 5 pyramid levels), "
                                    "MSCKF update of 70 features x 15 clones on n=113 (k=98 columns), points only"),
                       "replicas": world, "tracked_points": state["tracked"], "accepted_features": state["accepted"],
                       "lines_detected": state.get("lines_detected"), "lines_kept": state.get("lines_kept"), "accepted_lines": state.get("accepted_lines"), "points_only": points_only,
                       "lk_iterations_per_frame": int(state["lk_iters"]),
                       "front_end_arithmetic": "u8/int16/int64 exact + f32 2x2 solve", "update_arithmetic": "f64",
                       "schedule": "sequential, one stream" if args.sequential else
                                   "update of frame i overlapped with the front-end of frame i+1 (two contexts / streams)",
                       "sequential_ms_per_frame": seq_ms},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if uctx is not ctx:
        uctx.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
