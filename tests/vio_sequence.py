"""End-to-end synthetic visual-inertial sequence: the accuracy half of the metric (BASELINE.json: "ATE vs CPU reference").

A simulated platform moves along a smooth path past random landmarks; a noisy 200 Hz IMU and a 10 Hz camera (pixel noise)
feed a sliding-window MSCKF whose every numeric step goes through ONE backend:

  HipBackend     the C-ABI library on the GPU  (propagate, clone, marginalize, triangulate, Jacobians, MSCKF update)
  OracleBackend  the CPU restatement in oracle/ (the checker; never imported by the product)

The bookkeeping around those steps (which feature is used when, applying dx to the state vector, the clone list) is this
file's and is shared by both runs, so the two trajectories differ only by the arithmetic of the backends.  Trajectories are
written with plv_traj_format, read back with plv_traj_load, associated and scored with plv_traj_ate.

    python tests/vio_sequence.py [--seconds 20] [--out profiles/r01/sequence_ate.json]     (needs a GPU)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

G = np.array([0.0, 0.0, 9.81])
MAX_CLONES, MAX_MSCKF, CAM_HZ, IMU_HZ = 11, 40, 10.0, 200.0
SIGMA_PIX = 1.0   # the filter's measurement sigma
NOISE_PIX = 0.2   # simulated tracking noise: the reference only uses a feature whose whole residual has norm < 3 (UpdaterCamera.cpp:238)


# ----------------------------------------------------------------------------------------------- small SO(3) helpers
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def exp_so3(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3) + skew(w)
    K = skew(w / th)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def quat_2_rot(q):
    return (2 * q[3] ** 2 - 1) * np.eye(3) - 2 * q[3] * skew(q[:3]) + 2 * np.outer(q[:3], q[:3])


def rot_2_quat(R):
    T = np.trace(R)
    q = np.zeros(4)
    i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2], T]))
    if i == 3:
        q[3] = np.sqrt((1 + T) / 4)
        q[0], q[1], q[2] = (R[1, 2] - R[2, 1]) / (4 * q[3]), (R[2, 0] - R[0, 2]) / (4 * q[3]), (R[0, 1] - R[1, 0]) / (4 * q[3])
    else:
        j, k = (i + 1) % 3, (i + 2) % 3
        q[i] = np.sqrt((1 + 2 * R[i, i] - T) / 4)
        q[j], q[k] = (R[i, j] + R[j, i]) / (4 * q[i]), (R[i, k] + R[k, i]) / (4 * q[i])
        q[3] = (R[j, k] - R[k, j]) / (4 * q[i])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_left_update(q, dth):
    """JPLQuat::update: q <- quatnorm([dth / 2, 1]) (x) q  (REF: open_vins/ov_core/src/types/JPLQuat.h update())."""
    dq = np.array([0.5 * dth[0], 0.5 * dth[1], 0.5 * dth[2], 1.0])
    dq /= np.linalg.norm(dq)
    Qm = np.zeros((4, 4))
    Qm[:3, :3] = dq[3] * np.eye(3) - skew(dq[:3])
    Qm[:3, 3], Qm[3, :3], Qm[3, 3] = dq[:3], -dq[:3], dq[3]
    r = Qm @ q
    if r[3] < 0:
        r = -r
    return r / np.linalg.norm(r)


# ----------------------------------------------------------------------------------------------------- the world
def trajectory(t):
    """R_GtoI, p_IinG: a 1.2 m/s loop with gentle roll / pitch, camera looking along +x of the body."""
    s = t
    yaw = 0.25 * s
    R_ItoG = exp_so3(np.array([0, 0, yaw])) @ exp_so3(np.array([0.05 * np.sin(0.7 * s), 0.04 * np.cos(0.5 * s), 0.0]))
    p = np.array([4.8 * np.sin(0.25 * s), 4.8 * (1 - np.cos(0.25 * s)), 0.3 * np.sin(0.4 * s)])
    return R_ItoG.T, p


def make_world(seconds, seed):
    rng = np.random.default_rng(seed)
    K8 = synth.EUROC_K8.copy()
    # camera z axis = body x axis (forward), camera x = -body y, camera y = -body z
    R_ItoC = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]]) @ exp_so3(np.array([0.01, -0.02, 0.015]))
    p_IinC = np.array([0.05, -0.02, 0.01])
    pts = np.column_stack([rng.uniform(-16, 16, 6000), rng.uniform(-11, 21, 6000), rng.uniform(-4, 4, 6000)])
    t_imu, wm, am = synth.imu_stream(trajectory, 0.0, seconds + 0.05, rate=IMU_HZ)
    bg, ba = np.array([0.002, -0.003, 0.001]), np.array([0.02, 0.01, -0.015])
    sw, swb, sa, sab = 1.6968e-4, 1.9393e-5, 2.0e-3, 3.0e-3
    wm = wm + bg + rng.normal(0, sw * np.sqrt(IMU_HZ), wm.shape)
    am = am + ba + rng.normal(0, sa * np.sqrt(IMU_HZ), am.shape)
    frames = []
    for k in range(int(seconds * CAM_HZ) + 1):
        tk = k / CAM_HZ
        R, p = trajectory(tk)
        pc = (R_ItoC @ (R @ (pts - p).T)).T + p_IinC
        vis = np.flatnonzero((pc[:, 2] > 1.0) & (pc[:, 2] < 25.0))
        xn = pc[vis, :2] / pc[vis, 2:3]
        keep = np.hypot(xn[:, 0], xn[:, 1]) < 0.8
        vis, xn = vis[keep], xn[keep]
        uv = np.array([synth.radtan_distort(K8, x) for x in xn]).reshape(-1, 2)
        inb = (uv[:, 0] > 5) & (uv[:, 0] < 747) & (uv[:, 1] > 5) & (uv[:, 1] < 475)
        uv = uv[inb] + rng.normal(0, NOISE_PIX, (int(inb.sum()), 2))
        frames.append((tk, vis[inb], uv.astype(np.float32)))
    return dict(K8=K8, R_ItoC=R_ItoC, p_IinC=p_IinC, pts=pts, t_imu=t_imu, wm=wm, am=am, bg=bg, ba=ba, frames=frames,
                noise=(sw, swb, sa, sab))


# ----------------------------------------------------------------------------------------------------- backends
class HipBackend:
    name = "hip"

    def __init__(self, pkg):
        self.pkg = pkg
        self.ctx = pkg.Context(pkg.default_config(752, 480))
        self.n = 0

    def cov_set(self, P):
        self.ctx.cov_upload(P)
        self.n = P.shape[0]

    def cov_get(self):
        return self.ctx.cov_download(self.n)

    def propagate(self, imu, nz, t, wm, am):
        self.ctx.propagate(imu, nz, t, wm, am, self.n)

    def clone(self):
        self.ctx.cov_clone(self.n, 0, 6)
        self.n += 6

    def marginalize(self, idx, size):
        self.ctx.cov_marginalize(idx, size)
        self.n -= size

    def triangulate(self, st, tr):
        return self.ctx.triangulate(st, tr, max_cond=1e6, max_dist=60.0, max_baseline=1e3)

    def columns(self, st, tr):
        return self.ctx.jacobian_columns(st, tr)

    def update(self, st, tr, cols, ld, sigma2):
        self.ctx.build_jacobians_resident(st, tr, cols, ld)
        rc, dx, acc, nrows = self.ctx.msckf_update_resident(self.n, sigma2)
        return rc, dx, acc


class OracleBackend:
    name = "oracle"

    def __init__(self, pkg):
        import oracle_lib
        self.pkg = pkg
        self.o, self.jo, self.po = oracle_lib.load(), oracle_lib.load_jac(pkg), oracle_lib.load_prop(pkg)
        self.q95 = synth.q95_table()
        self.P = None

    def cov_set(self, P):
        self.P = np.array(P, order="F")

    def cov_get(self):
        return self.P

    def propagate(self, imu, nz, t, wm, am):
        self.P = self.po.propagate(imu, nz, t, wm, am, P=self.P)[3]

    def clone(self):
        self.P = self.po.cov_clone(self.P, 0, 6)

    def marginalize(self, idx, size):
        self.P = self.o.cov_marginalize(self.P, idx, size)

    def triangulate(self, st, tr):
        return self.jo.triangulate_batch(st, tr, max_cond=1e6, max_dist=60.0, max_baseline=1e3)

    def columns(self, st, tr):
        return self.jo.columns(st, tr)

    def update(self, st, tr, cols, ld, sigma2):
        rows, Hf, Hx, res = self.jo.build_jacobians(st, tr, cols, ld)
        rc, self.P, dx, acc, _ = self.o.msckf_update(self.P, rows, Hf, Hx, res, cols, sigma2, self.q95)
        return rc, dx, acc


# ------------------------------------------------------------------------------------------------------ the filter
def run_filter(pkg, backend, world, log=None):
    """Sliding-window MSCKF over the simulated sequence.  Returns (times, poses [n][7], stats)."""
    sw, swb, sa, sab = world["noise"]
    nz = pkg.imu_noise(sw, swb, sa, sab, tuple(G))
    R0, p0 = trajectory(0.0)
    v0 = (trajectory(1e-5)[1] - trajectory(-1e-5)[1]) / 2e-5
    rng = np.random.default_rng(99)
    # start from the truth perturbed inside the initial covariance
    imu = pkg.PlvImuState.make(rot_2_quat(exp_so3(rng.normal(0, 1e-3, 3)) @ R0), p0 + rng.normal(0, 1e-3, 3), v0 + rng.normal(0, 1e-2, 3),
                               world["bg"] * 0, world["ba"] * 0)
    K8 = world["K8"].copy()
    P = np.zeros((23, 23))
    P[np.arange(23), np.arange(23)] = [1e-6] * 3 + [1e-6] * 3 + [1e-4] * 3 + [1e-4] * 3 + [1e-2] * 3 + [1.0] * 4 + [1e-4] * 4
    backend.cov_set(P)
    clones = []        # dicts: t, R, p, Rf, pf
    tracks = {}        # track id -> dict(lm, t [], uv [])
    by_lm = {}         # landmark -> live track id
    next_id = 1
    t_state = 0.0
    out_t, out_pose = [], []
    n_used = n_acc = n_updates = 0

    def state_view():
        ids = 23 + 6 * np.arange(len(clones))
        return pkg.StateView([c["t"] for c in clones], [c["R"] for c in clones], [c["p"] for c in clones], ids, world["R_ItoC"],
                             world["p_IinC"], K8, clone_R_fej=[c["Rf"] for c in clones], clone_p_fej=[c["pf"] for c in clones],
                             intrinsic_state_id=15, sigma_pix=SIGMA_PIX)

    for tk, lms, uvs in world["frames"]:
        # ---- 1. propagate to the frame (Propagator::propagate) and clone (StateHelper::augment_clone)
        if tk > t_state:
            ok, st_, sw_, sa_ = pkg.select_imu_readings(world["t_imu"], world["wm"], world["am"], t_state, tk)
            assert ok
            backend.propagate(imu, nz, st_, sw_, sa_)
            t_state = tk
        backend.clone()
        R, p = quat_2_rot(np.array(imu.q)), np.array(imu.p)
        clones.append(dict(t=tk, R=R.copy(), p=p.copy(), Rf=R.copy(), pf=p.copy()))
        # ---- 2. observations
        seen = set()
        for lm, uv in zip(lms, uvs):
            lm = int(lm)
            tid = by_lm.get(lm)
            if tid is None:
                tid = next_id
                next_id += 1
                by_lm[lm] = tid
                tracks[tid] = dict(lm=lm, t=[], uv=[])
            tracks[tid]["t"].append(tk)
            tracks[tid]["uv"].append(uv)
            seen.add(tid)
        # ---- 3. features to use now: lost tracks and tracks as long as the window
        use = [tid for tid, tr in tracks.items() if (tid not in seen or len(tr["t"]) >= MAX_CLONES) and len(tr["t"]) >= 3]
        drop = [tid for tid, tr in tracks.items() if tid not in seen and len(tr["t"]) < 3]
        use.sort(key=lambda tid: (-len(tracks[tid]["t"]), tid))
        use = use[:MAX_MSCKF]
        if len(clones) >= 4 and use:
            st = state_view()
            ptr = np.concatenate([[0], np.cumsum([len(tracks[i]["t"]) for i in use])]).astype(np.int32)
            tt = np.concatenate([tracks[i]["t"] for i in use])
            uv = np.concatenate([tracks[i]["uv"] for i in use]).astype(np.float32)
            uvn = np.array([_undistort(K8, x) for x in uv], dtype=np.float32)
            tr_all = pkg.Tracks(ptr, tt, uv, np.zeros((len(use), 3)), obs_uvn=uvn)
            pf, okf, err = backend.triangulate(st, tr_all)
            sel = [q for q in range(len(use)) if okf[q] and err[q] < 3.0]
            if sel:
                sptr = np.concatenate([[0], np.cumsum([ptr[q + 1] - ptr[q] for q in sel])]).astype(np.int32)
                idx = np.concatenate([np.arange(ptr[q], ptr[q + 1]) for q in sel])
                tr = pkg.Tracks(sptr, tt[idx], uv[idx], pf[sel])
                cols = backend.columns(st, tr)
                rc, dx, acc = backend.update(st, tr, cols, 2 * (MAX_CLONES + 1), SIGMA_PIX ** 2)
                n_used += len(sel)
                n_acc += int(acc.sum())
                if rc == 0 and acc.any():
                    n_updates += 1
                    # ---- 4. x <- x [+] dx  (IMU::update, Vec::update, PoseJPL::update)
                    q = quat_left_update(np.array(imu.q), dx[0:3])
                    for i in range(4):
                        imu.q[i] = q[i]
                    for i in range(3):
                        imu.p[i] += dx[3 + i]
                        imu.v[i] += dx[6 + i]
                        imu.bg[i] += dx[9 + i]
                        imu.ba[i] += dx[12 + i]
                    K8 = K8 + dx[15:23]
                    for ci, c in enumerate(clones):
                        d = dx[23 + 6 * ci:29 + 6 * ci]
                        c["R"] = quat_2_rot(quat_left_update(rot_2_quat(c["R"]), d[:3]))
                        c["p"] = c["p"] + d[3:]
        for tid in use + drop:
            lm = tracks[tid]["lm"]
            if by_lm.get(lm) == tid:
                del by_lm[lm]
            del tracks[tid]
        # ---- 5. window maintenance (StateHelper::marginalize_old_clone)
        if len(clones) > MAX_CLONES:
            backend.marginalize(23, 6)
            told = clones.pop(0)["t"]
            for tid in list(tracks):
                tr = tracks[tid]
                if tr["t"] and tr["t"][0] <= told:
                    tr["t"], tr["uv"] = tr["t"][1:], tr["uv"][1:]
        out_t.append(tk)
        out_pose.append(np.concatenate([np.array(imu.p), np.array(imu.q)]))
        if log and len(out_t) % 20 == 0:
            Rt, pt = trajectory(tk)
            log(f"  [{backend.name}] t={tk:5.1f}s  |p - p_true| = {np.linalg.norm(np.array(imu.p) - pt):.3f} m   updates {n_updates}  accepted {n_acc}/{n_used}")
    return np.array(out_t), np.array(out_pose), dict(updates=n_updates, features_used=n_used, features_accepted=n_acc)


def _undistort(K8, uv, iters=8):
    """radtan inverse by fixed-point iteration (the harness' own; the product's undistort is exercised elsewhere)."""
    x = np.array([(uv[0] - K8[2]) / K8[0], (uv[1] - K8[3]) / K8[1]])
    x0 = x.copy()
    for _ in range(iters):
        r2 = x @ x
        rad = 1 + K8[4] * r2 + K8[5] * r2 * r2
        dx = np.array([2 * K8[6] * x[0] * x[1] + K8[7] * (r2 + 2 * x[0] ** 2), K8[6] * (r2 + 2 * x[1] ** 2) + 2 * K8[7] * x[0] * x[1]])
        x = (x0 - dx) / rad
    return x


def ground_truth(times):
    out = np.zeros((len(times), 7))
    for i, t in enumerate(times):
        R, p = trajectory(t)
        out[i, :3], out[i, 3:] = p, rot_2_quat(R)
    return out


def write_and_reload(pkg, path, times, poses):
    with open(path, "w") as f:
        f.write(pkg.traj_header())
        for t, ps in zip(times, poses):
            f.write(pkg.traj_format(t, ps[:3], ps[3:]))
    return pkg.traj_load(path)[:2]


def evaluate(pkg, ctx, workdir, runs, method="posyaw"):
    """runs: name -> (times, poses).  ATE of every run against the ground truth and of hip against oracle, through the file path."""
    res = {}
    loaded = {name: write_and_reload(pkg, os.path.join(workdir, f"traj_{name}.txt"), t, p) for name, (t, p) in runs.items()}
    any_t = next(iter(loaded.values()))[0]
    gt_t, gt_p = write_and_reload(pkg, os.path.join(workdir, "traj_gt.txt"), any_t, ground_truth(any_t))
    for name, (t, p) in loaded.items():
        ei, gi = pkg.traj_associate(t, gt_t)
        r = ctx.traj_ate(p[ei], gt_p[gi], method)
        res[f"{name}_vs_truth"] = dict(pos=r["pos"], ori=r["ori"], n=len(ei), length_m=pkg.traj_length(p))
    if "hip" in loaded and "oracle" in loaded:
        ei, gi = pkg.traj_associate(loaded["hip"][0], loaded["oracle"][0])
        r = ctx.traj_ate(loaded["hip"][1][ei], loaded["oracle"][1][gi], "none")
        res["hip_vs_oracle_logged"] = dict(pos=r["pos"], ori=r["ori"], n=len(ei))   # 6-decimal log files
        d = runs["hip"][1] - runs["oracle"][1]
        res["hip_vs_oracle_raw"] = dict(max_pos_diff_m=float(np.abs(d[:, :3]).max()), max_quat_diff=float(np.abs(d[:, 3:]).max()))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--backend", default="both", choices=["both", "hip", "oracle"])
    args = ap.parse_args()
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    world = make_world(args.seconds, args.seed)
    runs, stats, wall = {}, {}, {}
    for B in [b for b in (HipBackend, OracleBackend) if args.backend in ("both", b.name)]:
        b = B(pkg)
        t0 = time.time()
        t, p, s = run_filter(pkg, b, world, log=print)
        wall[b.name] = time.time() - t0
        runs[b.name], stats[b.name] = (t, p), s
    if args.backend == "oracle":  # CPU-only dry run of the bookkeeping: no device, no score
        print(json.dumps(dict(filter=stats, wall_s=wall)))
        return
    ctx = pkg.Context(pkg.default_config(752, 480))
    with tempfile.TemporaryDirectory() as d:
        res = evaluate(pkg, ctx, d, runs)
    res["filter"] = stats
    res["wall_s"] = wall
    res["config"] = dict(seconds=args.seconds, cam_hz=CAM_HZ, imu_hz=IMU_HZ, max_clones=MAX_CLONES, max_msckf=MAX_MSCKF, sigma_pix=SIGMA_PIX, noise_pix=NOISE_PIX,
                         seed=args.seed, frames=len(world["frames"]))
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
