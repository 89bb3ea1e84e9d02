"""A synthetic dataset on disk in the layout pl-viwo_amd/replay.py reads (test infrastructure): a wheeled vehicle drives a circle on a
textured ground inside a square room with textured walls; a forward-looking radtan camera is rendered by ray casting (per-pixel rays
undistorted once, mip-mapped texture lookup), the IMU is sampled from the analytic trajectory, the wheel encoders from the unicycle
model.  Also writes the configuration directory (the reference's YAML layout) and the ground-truth trajectory.

    python tests/synth_dataset.py OUT_DIR [--seconds 12]
"""
import argparse
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

G = np.array([0.0, 0.0, 9.81])
W, H = 752, 480
K8 = synth.EUROC_K8.copy()


def set_camera(width=752, height=480):
    """Image size of the rendered camera (module-wide, like PATH / REST): 752x480 with the EuRoC intrinsics, or those intrinsics
    scaled to another width (BASELINE configs[3]: 1280x720)."""
    global W, H, K8
    W, H = int(width), int(height)
    K8 = synth.EUROC_K8.copy()
    if (W, H) != (752, 480):
        sx = W / 752.0
        K8[:4] = K8[0] * sx, K8[1] * sx, W / 2.0 + 7.0, H / 2.0 + 8.5
RADIUS, WALL_R, WALL_H = 8.0, 21.0, 9.0
BOULEVARD = dict(wall_r=6.5)   # "boulevard" style: overrides of _mips_boulevard's bar geometry (texels of 0.012 m)
RL, RR, BASE = 0.31, 0.305, 1.52
# IMU: mounted flat (z up), 0.9 m above the odometry frame, slightly yawed.  wheel_extrinsic = (R_ItoO, p_IinO)
R_ITOO = Rotation.from_rotvec([0.0, 0.0, 0.02]).as_matrix()
P_IINO = np.array([0.35, -0.05, 0.9])
# camera: z forward (body x), x right (-body y), y down (-body z), pitched 12 deg down, 0.4 m ahead of and above the IMU
R_CTOI = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]]) @ Rotation.from_rotvec([np.deg2rad(12.0), 0, 0]).as_matrix()
P_CINI = np.array([0.4, 0.02, 0.35])


def set_mount(pitch_down_deg=12.0, yaw_right_deg=0.0):
    """Camera mount (module-wide, like set_camera): pitched down and yawed to the right of the driving direction."""
    global R_CTOI
    R_CTOI = (np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]]) @ Rotation.from_rotvec([0, np.deg2rad(yaw_right_deg), 0]).as_matrix()
              @ Rotation.from_rotvec([np.deg2rad(pitch_down_deg), 0, 0]).as_matrix())
BG, BA = np.array([0.003, -0.002, 0.0015]), np.array([0.03, -0.02, 0.025])
SIG = dict(gyro_noise=1.7e-4, gyro_bias=1.9e-5, accel_noise=2.0e-3, accel_bias=3.0e-3)


REST = 0.0   # seconds at rest before the vehicle pulls away (make_dataset(rest=...)); 0 = moving from the first sample


def arc(t):
    """arc length and speed along the circle.  REST = 0: 2.6 m/s with a slow modulation, never at rest (the IMU-wheel initialiser
    runs its dynamic branch).  REST > 0: standing still, then pulling away with 2.6 m/s^2 towards 2.6 m/s (the jerk the static IMU
    initialiser waits for)."""
    if REST > 0:
        x = np.maximum(t - REST, 0.0)
        return 2.6 * (x - (1 - np.exp(-x))), 2.6 * (1 - np.exp(-x))
    return 2.6 * t + 0.8 * np.sin(0.7 * t) / 0.7, 2.6 + 0.8 * np.cos(0.7 * t)


PATH = "circle"   # "circle": radius-8 m circle (the replay tests); "street": straight down a corridor along +x with a slight weave
WEAVE_A, WEAVE_K = 0.25, 2 * np.pi / 45.0


def odom_pose(t):
    s, _ = arc(t)
    if PATH == "street":
        # the reference classifies lines by the vanishing points of the IMU axes (LineHelper.cpp:1026-1088) and triangulates a
        # classified line from one of its points and that axis (:226-293): structure along / across the driving direction, as on a road
        th = np.arctan(WEAVE_A * WEAVE_K * np.cos(WEAVE_K * s))
        R_OtoG = Rotation.from_rotvec([0, 0, th]).as_matrix()
        return R_OtoG.T, np.array([s, WEAVE_A * np.sin(WEAVE_K * s), 0.0])
    th = s / RADIUS
    R_OtoG = Rotation.from_rotvec([0, 0, th]).as_matrix()
    return R_OtoG.T, np.array([RADIUS * np.sin(th), RADIUS * (1 - np.cos(th)), 0.0])


def twist(t, h=1e-4):
    """speed along the path and yaw rate of the odometry frame"""
    if PATH != "street":
        v = arc(t)[1]
        return v, v / RADIUS
    (Rm, pm), (Rp, pp) = odom_pose(t - h), odom_pose(t + h)
    dth = Rotation.from_matrix(Rp.T @ Rm).as_rotvec()[2]
    return np.linalg.norm(pp - pm) / (2 * h), dth / (2 * h)


def imu_pose(t):
    R_GtoO, p_O = odom_pose(t)
    R_GtoI = R_ITOO.T @ R_GtoO
    return R_GtoI, p_O + R_GtoI.T @ (R_ITOO.T @ P_IINO)


def rot_2_quat(R):
    q = Rotation.from_matrix(R.T).as_quat()   # JPL q_GtoI has the Hamilton components of R_ItoG
    return -q if q[3] < 0 else q


# --------------------------------------------------------------------------------------------------------------- rendering
def _undistorted_rays():
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    x0, y0 = (xs - K8[2]) / K8[0], (ys - K8[3]) / K8[1]
    x, y = x0.copy(), y0.copy()
    for _ in range(12):
        r2 = x * x + y * y
        rad = 1 + K8[4] * r2 + K8[5] * r2 * r2
        dx = 2 * K8[6] * x * y + K8[7] * (r2 + 2 * x * x)
        dy = K8[6] * (r2 + 2 * y * y) + 2 * K8[7] * x * y
        x, y = (x0 - dx) / rad, (y0 - dy) / rad
    d = np.stack([x, y, np.ones_like(x)], axis=-1)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def _mips(seed, manhattan=True):
    """texture + its mip pyramid; with `manhattan` the bright straight edges run along the two texture axes only (on the walls:
    horizontal and vertical structural lines, the kind the line front-end classifies by vanishing point)"""
    base = synth.texture_canvas(2048 - 128, 2048 - 128, seed=seed, margin=64, blobs=9000, lines=0 if manhattan else 120)
    if manhattan:
        rng = np.random.default_rng(seed + 100)
        n = base.shape[0]
        for _ in range(140):
            L, c0, c1 = int(rng.uniform(80, 400)), int(rng.uniform(8, n - 8)), int(rng.uniform(0, n - 400))
            if rng.random() < 0.5:
                base[c0 - 1:c0 + 2, c1:c1 + L] = 1.0
            else:
                base[c1:c1 + L, c0 - 1:c0 + 2] = 1.0
    out = [base]
    while out[-1].shape[0] > 16:
        a = out[-1]
        out.append(0.25 * (a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2]))
    return out


def _mips_street(seed, ground=False, dense=False):
    """"street" style texture (the bench / line-test scene): a smooth large-scale shading with little fine texture, covered with
    axis-aligned high-contrast rectangles (facade panels and windows on the walls, paving slabs on the ground): long straight edges that
    end in corners, so that FAST corners sit on the segments the line detector finds (TrackLSD keeps a line only when a tracked point lies
    on it, TrackLSD.cpp:744-792)."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    n = 2048 - 128
    base = ndi.gaussian_filter(rng.normal(0, 1, (n, n)), 48.0, mode="wrap")
    base = 0.5 + 0.22 * base / np.abs(base).max() + 0.018 * ndi.gaussian_filter(rng.normal(0, 1, (n, n)), 1.2, mode="wrap") / 0.2
    if ground:
        tile = 64 if dense else 120                 # 1.4 m slabs (dense: 0.77 m), every one its own grey level
        g = rng.uniform(0.12, 0.88, (n // tile + 1, n // tile + 1))
        base = 0.35 * base + 0.65 * np.kron(g, np.ones((tile, tile)))[:n, :n]
    else:
        def rect(x, y, w, h):
            v = rng.uniform(0.04, 0.3) if rng.random() < 0.5 else rng.uniform(0.7, 0.96)
            base[y:y + h, x:x + w] = 0.15 * base[y:y + h, x:x + w] + 0.85 * v

        for _ in range(260 if dense else 110):      # panels 1.7 .. 5.5 m wide (dense: more and smaller ones, 0.8 .. 4 m)
            w, h = (int(rng.uniform(70, 330)), int(rng.uniform(60, 220))) if dense else (int(rng.uniform(140, 460)), int(rng.uniform(90, 300)))
            rect(int(rng.uniform(0, n - w)), int(rng.uniform(0, n - h)), w, h)
        for _ in range(90 if dense else 40):        # window grids: rows of 0.7 x 1.1 m openings
            w, h, gx, gy = int(rng.uniform(50, 75)), int(rng.uniform(80, 110)), int(rng.integers(2, 6)), int(rng.integers(1, 4))
            px, py = int(w * rng.uniform(1.5, 2.2)), int(h * rng.uniform(1.3, 1.8))
            x0, y0 = int(rng.uniform(0, n - gx * px)), int(rng.uniform(0, n - gy * py))
            for i in range(gx):
                for j in range(gy):
                    rect(x0 + i * px, y0 + j * py, w, h)
    base = np.clip(ndi.gaussian_filter(base, 0.7, mode="wrap"), 0, 1)
    out = [base]
    while out[-1].shape[0] > 16:
        a = out[-1]
        out.append(0.25 * (a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2]))
    return out


def _mips_boulevard(seed, plain=False, angle=24.0, pitch=16.0, cell=104.0, joint=9.0, zig=4.2, period=(13.0, 21.0), **_):
    """"boulevard" style wall texture (the round-5 bench scene, BASELINE configs[2] with the metric's 80 kept lines): slanted courses of
    two-tone strips (`pitch` texels = 0.19 m wide, `angle` degrees off the horizontal; upper half bright, lower half dark, the boundary
    between the halves an irregular zigzag), cut into ~1.2 m pieces by narrow joints of the background tone, each piece with its own
    offset across the slant.  The straight edges are the course boundaries (dark above, bright below, all stepping the same way); the
    zigzag boundary yields no straight segment, and its teeth put corners within 5 px of the course boundaries, away from their ends.
    Why: FastLineDetector orients a segment by the brightness of its two sides, and TrackLSD::AssignPointToLines' bounding-box test
    (with its end-point mix-up, TrackLSD.cpp:754-765) passes a point of a segment only when the segment runs up-left in the image's
    x > y part (down-right in the x < y part) and the point is not beyond the segment's ends.  With random polarity and mostly
    axis-parallel edges two thirds of the segments that own a tracked point are dropped ('avenue': 190 detected, 95 with a point, 32
    kept per frame); here the side-looking camera of the boulevard drive sees every course boundary run up-left with the darker side
    on its upper right.  `plain`: the background only (ground and ceiling)."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    n = 2048 - 128
    base = ndi.gaussian_filter(rng.normal(0, 1, (n, n)), 48.0, mode="wrap")
    base = 0.5 + 0.05 * base / np.abs(base).max() + 0.003 * ndi.gaussian_filter(rng.normal(0, 1, (n, n)), 1.2, mode="wrap") / 0.2
    if not plain:
        th = np.deg2rad(angle)
        X, Y = np.meshgrid(np.arange(n) + 0.5, np.arange(n) + 0.5)       # texture u (columns), v (rows; up on a wall)
        al, ac = X * np.cos(th) + Y * np.sin(th), -X * np.sin(th) + Y * np.cos(th)
        ci = np.floor(al / cell).astype(int)
        c0, ncell = ci.min(), ci.max() - ci.min() + 1
        ac = ac + rng.uniform(0, pitch, ncell)[ci - c0]                    # every piece column has its own offset across the slant
        row = np.floor(ac / pitch).astype(int)
        bb = ac - row * pitch
        r0, nr = row.min(), row.max() - row.min() + 1
        key = (row - r0) * ncell + (ci - c0)                                # one strip piece
        T = rng.uniform(*period, nr * ncell)[key]
        ph = rng.uniform(0, 1.0, nr * ncell)[key]
        hi, lo = (0.5 + rng.uniform(0.3, 0.42, nr * ncell))[key], (0.5 - rng.uniform(0.3, 0.42, nr * ncell))[key]
        x = al / T + ph + 0.35 * np.sin(al / 57.0 + 6.28 * ph)
        tri = 2.0 * np.abs(2.0 * (x - np.floor(x + 0.5))) - 1.0           # triangle wave in [-1, 1]
        strip = np.where(bb > 0.5 * pitch + zig * tri, hi, lo)
        base = np.where(al - ci * cell < joint, base, strip)
    base = np.clip(ndi.gaussian_filter(base, 0.7, mode="wrap"), 0, 1)
    out = [base]
    while out[-1].shape[0] > 16:
        a = out[-1]
        out.append(0.25 * (a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2]))
    return out


def _sample(mips, u, v, level):
    """bilinear lookup of texel coordinates (u, v) of level 0 at the mip level chosen per pixel, wrapping around"""
    out = np.zeros(u.shape)
    lv = np.clip(np.rint(level), 0, len(mips) - 1).astype(int)
    for l in np.unique(lv):
        m = lv == l
        tex = mips[l]
        n = tex.shape[0]
        uu, vv = u[m] / (1 << l) - 0.5, v[m] / (1 << l) - 0.5
        i0, j0 = np.floor(uu).astype(int), np.floor(vv).astype(int)
        fu, fv = uu - i0, vv - j0
        i0, j0, i1, j1 = i0 % n, j0 % n, (i0 + 1) % n, (j0 + 1) % n
        out[m] = (1 - fv) * ((1 - fu) * tex[j0, i0] + fu * tex[j0, i1]) + fv * ((1 - fu) * tex[j1, i0] + fu * tex[j1, i1])
    return out


class Renderer:
    TEXEL = 0.012   # metres per level-0 texel

    def __init__(self, seed=7, style="room"):
        """style "room": noise-textured walls 21 m out with thin bright lines (the replay tests' scene); "street": a corridor with facades 9 m to either side with
        panels / windows and a paved ground (_mips_street), 7 m high with a ceiling instead of the empty sky"""
        self.rays = _undistorted_rays().reshape(-1, 3)
        self.style = style
        self.mirror = False
        if style in ("street", "avenue", "boulevard"):   # "avenue": the street with twice the structure (the round-3 bench scene)
            dense = style == "avenue"
            if style == "boulevard":                # slanted ramp-shaded bars on the walls, plain ground and ceiling (_mips_boulevard)
                self.ground, self.wall = _mips_boulevard(seed, plain=True), _mips_boulevard(seed + 1, **BOULEVARD)
            else:
                self.ground, self.wall = _mips_street(seed, ground=True, dense=dense), _mips_street(seed + 1, dense=dense)
            self.wall_r, self.wall_h = (BOULEVARD.get("wall_r", 9.0), 7.0) if style == "boulevard" else (9.0, 7.0)
            self.style = "street"
        else:
            self.ground, self.wall = _mips(seed), _mips(seed + 1)
            self.wall_r, self.wall_h = WALL_R, WALL_H
        self.f = 0.5 * (K8[0] + K8[1])

    def render(self, t):
        R_GtoI, p_I = imu_pose(t)
        R_CtoG = R_GtoI.T @ R_CTOI
        c = p_I + R_GtoI.T @ P_CINI
        d = self.rays @ R_CtoG.T
        img = np.full(len(d), 0.55)
        # ground z = 0
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(d[:, 2] < -1e-6, -c[2] / d[:, 2], np.inf)
        # four planar walls of a square room (half side WALL_R) centred on the circle: lines drawn on them are straight in 3D
        best_t = np.where(np.isfinite(tg), tg, np.inf)
        img_hit = np.full(len(d), -1)            # -1 sky, 0 ground, 1.. wall index + 1
        img_hit[np.isfinite(tg)] = 0
        if self.style == "street":   # a corridor along x: side walls at y = +-wall_r, end walls far away
            ox, oy = c[0] - 50.0, c[1]
            plan = ((0, 1.0, 150.0, self.wall_r), (0, -1.0, 150.0, self.wall_r), (1, 1.0, self.wall_r, 150.0), (1, -1.0, self.wall_r, 150.0))
        else:                        # a square room centred on the circle
            ox, oy = c[0], c[1] - RADIUS
            plan = tuple((axis, sign, self.wall_r, self.wall_r) for axis, sign in ((0, 1.0), (0, -1.0), (1, 1.0), (1, -1.0)))
        walls = []
        for k, (axis, sign, dist_w, half_w) in enumerate(plan):
            o, dd = (ox, d[:, 0]) if axis == 0 else (oy, d[:, 1])
            with np.errstate(divide="ignore", invalid="ignore"):
                tw = np.where(sign * dd > 1e-9, (sign * dist_w - o) / dd, np.inf)
            zw = c[2] + tw * d[:, 2]
            other = (oy + tw * d[:, 1]) if axis == 0 else (ox + tw * d[:, 0])
            ok = np.isfinite(tw) & (zw >= 0) & (zw <= self.wall_h) & (np.abs(other) <= half_w)
            tw = np.where(ok, tw, np.inf)
            closer = tw < best_t
            best_t = np.where(closer, tw, best_t)
            img_hit[closer] = k + 1
            walls.append((axis, sign))
        if self.style == "street":               # ceiling z = wall_h, paved like the ground (shifted)
            with np.errstate(divide="ignore", invalid="ignore"):
                tc = np.where(d[:, 2] > 1e-6, (self.wall_h - c[2]) / d[:, 2], np.inf)
            closer = tc < best_t
            best_t = np.where(closer, tc, best_t)
            img_hit[closer] = 5
        for which in range(0, 6):
            hit = img_hit == which
            if not hit.any():
                continue
            dist = best_t[hit]
            p = c + dist[:, None] * d[hit]
            if which == 0 or which == 5:
                mips = self.ground
                off = 0.0 if which == 0 else 777.0
                u, v = p[:, 0] / self.TEXEL + off, p[:, 1] / self.TEXEL + 2 * off
                if self.mirror and which == 5:
                    u = -u
                cosi = np.abs(d[hit, 2])
            else:
                axis, sign = walls[which - 1]
                mips = self.wall
                along = (p[:, 1] - (0.0 if self.style == "street" else RADIUS)) if axis == 0 else p[:, 0]
                u, v = (along + 2.0 * self.wall_r * which) / self.TEXEL, p[:, 2] / self.TEXEL
                if self.mirror and sign < 0:
                    u = -u
                cosi = np.abs(d[hit, axis])
            foot = dist / self.f / np.maximum(cosi, 0.15)
            level = np.log2(np.maximum(foot / self.TEXEL, 1.0))
            img[hit] = _sample(mips, u, v, level)
        return np.clip(np.rint(img.reshape(H, W) * 255.0), 0, 255).astype(np.uint8)


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.uint8).tobytes())


# ------------------------------------------------------------------------------------------------------------------ config
def write_config(cfg_dir, dataset_dir, traj_path, use_wheel=True, use_lines=True, aligned=True, clone_freq=10, n_pts=250, max_msckf=60,
                 calib_int=False, sigma_px=1.0):
    os.makedirs(cfg_dir, exist_ok=True)
    T_ic = np.eye(4)
    T_ic[:3, :3], T_ic[:3, 3] = R_CTOI, P_CINI
    # T_imu_wheel = [R_OtoI  p_OinI]
    T_iw = np.eye(4)
    T_iw[:3, :3], T_iw[:3, 3] = R_ITOO.T, -R_ITOO.T @ P_IINO
    mat = lambda T: "\n".join("    - [" + ", ".join(f"{x:.12g}" for x in row) + "]" for row in T)
    files = {
        "config.yaml": "%YAML:1.0\n\n" + "\n".join(f'config_{k}: "config_{k}.yaml"' for k in ("system", "estimator", "camera", "imu", "wheel", "init")) + "\n",
        "config_system.yaml": f'''%YAML:1.0

sys:
  verbosity: 2
  save_timing: false
  path_timing: "{os.path.join(os.path.dirname(traj_path), "timing.txt")}"
  save_state: false
  path_state: "{os.path.dirname(traj_path)}"
  save_trajectory: true
  path_trajectory: "{traj_path}"
  save_prints: false
  exp_id: 0
  path_bag: "{dataset_dir}"
  bag_start: 0
  bag_durr: -1
''',
        "config_estimator.yaml": f'''%YAML:1.0

est:
  gravity_mag: 9.81
  clone_freq: {clone_freq}
  window_size: 1.0
  intr_order: 3
  intr_error_mlt: 3
  intr_error_ori_thr: 0.007
  intr_error_pos_thr: 0.003
  intr_error_thr_mlt: 0.5
  dt_extrapolation: 0.01
  use_imu_res: false
  use_imu_cov: false
  use_pol_cov: true
  dynamic_cloning: false

intr_ori:
  Hz_10: [0.00288, 0.00126, 0.00108, 0.00102, 0.00102]
  Hz_15: [0.00138, 0.00066, 0.00063, 0.00069, 0.00087]
  Hz_20: [0.00084, 0.00012, 0.00006, 0.00003, 0.00003]
intr_pos:
  Hz_10: [0.00312, 0.00087, 0.00072, 0.00066, 0.00066]
  Hz_15: [0.00144, 0.00021, 0.00018, 0.00015, 0.00015]
  Hz_20: [0.00084, 0.00009, 0.00006, 0.00003, 0.00003]
''',
        "config_camera.yaml": f'''%YAML:1.0

cam:
  enabled: true
  max_n: 1
  use_stereo: false
  do_calib_ext: false
  do_calib_int: {'true' if calib_int else 'false'}
  do_calib_dt: false
  n_pts: {n_pts}
  fast: 20
  grid_x: 5
  grid_y: 5
  min_px_dist: 10
  knn: 0.85
  downsample: false
  histogram_method: "HISTOGRAM"
  max_slam: 0
  max_msckf: {max_msckf}
  feat_rep: "GLOBAL_3D"
  init_cov_dt: 1e-4
  init_cov_ex_o: 1e-4
  init_cov_ex_p: 1e-4
  init_cov_in_k: 1e-2
  init_cov_in_c: 1e-1
  init_cov_in_r: 1e-6
  sigma_px: {sigma_px}
  chi2_mult: 1
  fi_max_dist: 80
  fi_max_baseline: 2000
  fi_max_cond_number: 30000

cam0:
  timeoffset: 0.0
  T_imu_cam:
{mat(T_ic)}
  distortion_coeffs: [{", ".join(f"{x:.10g}" for x in K8[4:])}]
  distortion_model: radtan
  intrinsics: [{", ".join(f"{x:.10g}" for x in K8[:4])}]
  resolution: [{W}, {H}]
  topic: "cam0"
''',
        "config_imu.yaml": f'''%YAML:1.0

imu:
  accel_noise: {SIG["accel_noise"]}
  accel_bias: {SIG["accel_bias"]}
  gyro_noise: {SIG["gyro_noise"]}
  gyro_bias: {SIG["gyro_bias"]}
  topic: "imu0"
''',
        "config_wheel.yaml": f'''%YAML:1.0

wheel:
  enabled: {"true" if use_wheel else "false"}
  type: "Wheel3DAng"
  do_calib_dt: false
  do_calib_ext: false
  do_calib_int: false
  noise_w: 0.02
  noise_v: 0.02
  noise_p: 0.05
  init_cov_dt: 1e-4
  init_cov_ex_o: 1e-4
  init_cov_ex_p: 1e-3
  init_cov_in_b: 1e-3
  init_cov_in_r: 1e-3
  chi2_mult: 2
  timeoffset: 0.0
  intrinsics: [{RL}, {RR}, {BASE}]
  reuse_of_information: false
  T_imu_wheel:
{mat(T_iw)}
  topic: "wheel"
''',
        "config_init.yaml": f'''%YAML:1.0

init:
  window_time: 1.0
  imu_thresh: 0.3
  imu_wheel_thresh: 0.1
  imu_only_init: false
  imu_gravity_aligned: {"true" if aligned else "false"}
  use_gt: false
  cov_size: 1e-3
''',
    }
    for name, text in files.items():
        with open(os.path.join(cfg_dir, name), "w") as f:
            f.write(text)
    return os.path.join(cfg_dir, "config.yaml")


# ----------------------------------------------------------------------------------------------------------------- dataset
CAM_PHASE = float(os.environ.get("PLV_SYNTH_CAM_PHASE", "0.0017"))


def simulate(seconds=12.0, cam_hz=10.0, imu_hz=200.0, wheel_hz=50.0, seed=0, rest=0.0, style="room"):
    """The sensor streams of the synthetic drive, in memory: imu (t, wm, am), wheel (t, m1, m2), cam_times, gt (t, p, q)."""
    global REST, PATH
    REST = float(rest)
    PATH = "street" if style in ("street", "avenue", "boulevard") else "circle"
    rng = np.random.default_rng(seed)
    t, wm, am = synth.imu_stream(imu_pose, 0.0, seconds + 0.1, rate=imu_hz, bg=BG, ba=BA)
    wm = wm + rng.normal(0, SIG["gyro_noise"] * np.sqrt(imu_hz), wm.shape)
    am = am + rng.normal(0, SIG["accel_noise"] * np.sqrt(imu_hz), am.shape)
    tw = 0.0031 + np.arange(int((seconds + 0.1) * wheel_hz)) / wheel_hz
    m = np.zeros((len(tw), 2))
    for i, x in enumerate(tw):
        v, w = twist(x)
        m[i] = (v - w * BASE / 2) / RL + rng.normal(0, 0.02), (v + w * BASE / 2) / RR + rng.normal(0, 0.02)
    # (the camera is not synchronised with the IMU, as on the KAIST car: a frame stamped on an IMU sample's instant makes the clone
    # taken for it and the IMU pose of the same instant two window poses 1e-15 s apart whenever the clone time was formed as
    # "last clone + 1 / rate" — the interpolation polynomial through such a pair is undefined, in the reference as much as here)
    tc = 0.05 + CAM_PHASE + np.arange(int(seconds * cam_hz)) / cam_hz
    gt = []
    for x in t[::2]:
        R, p = imu_pose(x)
        gt.append(np.concatenate([[x], p, rot_2_quat(R)]))
    return dict(imu=(t, wm, am), wheel=(tw, m[:, 0], m[:, 1]), cam_times=tc, gt=np.array(gt), style=style)


_POOL_RD = None


def _render_one(x):
    return _POOL_RD.render(x)


def render_frames(times, style="room", workers=1, seed=7):
    """The camera images at `times`; workers > 1 renders in forked processes (call before anything initialises the GPU)."""
    global _POOL_RD, PATH
    PATH = "street" if style in ("street", "avenue", "boulevard") else "circle"
    _POOL_RD = Renderer(seed=seed, style=style)
    if workers <= 1 or len(times) < 4:
        return [_POOL_RD.render(x) for x in times]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(workers, len(times))) as pool:
        return pool.map(_render_one, list(times), chunksize=1)


def make_dataset(out_dir, seconds=12.0, cam_hz=10.0, imu_hz=200.0, wheel_hz=50.0, seed=0, render=True, log=None, rest=0.0, style="room",
                 workers=1):
    sim = simulate(seconds, cam_hz, imu_hz, wheel_hz, seed, rest, style)
    os.makedirs(os.path.join(out_dir, "cam0", "data"), exist_ok=True)
    t, wm, am = sim["imu"]
    with open(os.path.join(out_dir, "imu.csv"), "w") as f:
        f.write("# t wx wy wz ax ay az\n")
        for i in range(len(t)):
            f.write(f"{t[i]:.9f},{wm[i, 0]:.12g},{wm[i, 1]:.12g},{wm[i, 2]:.12g},{am[i, 0]:.12g},{am[i, 1]:.12g},{am[i, 2]:.12g}\n")
    tw, m1, m2 = sim["wheel"]
    with open(os.path.join(out_dir, "wheel.csv"), "w") as f:
        f.write("# t left right (wheel angular velocities, rad/s)\n")
        for i in range(len(tw)):
            f.write(f"{tw[i]:.9f},{m1[i]:.12g},{m2[i]:.12g}\n")
    tc = sim["cam_times"]
    imgs = None
    if render and workers > 1:
        imgs = render_frames(tc, style, workers)
    rd = Renderer(style=style) if render and imgs is None else None
    with open(os.path.join(out_dir, "cam0", "data.csv"), "w") as f:
        f.write("# t file\n")
        for k, x in enumerate(tc):
            name = f"{k:06d}.pgm"
            f.write(f"{x:.9f},{name}\n")
            if render:
                write_pgm(os.path.join(out_dir, "cam0", "data", name), imgs[k] if imgs is not None else rd.render(x))
            if log and k % 20 == 0:
                log(f"rendered {k}/{len(tc)}")
    with open(os.path.join(out_dir, "gt.txt"), "w") as f:
        f.write("# timestamp(s) tx ty tz qx qy qz qw\n")
        for g in sim["gt"]:
            f.write(f"{g[0]:.6f} {g[1]:.6f} {g[2]:.6f} {g[3]:.6f} {g[4]:.8f} {g[5]:.8f} {g[6]:.8f} {g[7]:.8f}\n")
    return dict(imu=(t, wm, am), cam_times=tc)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--seconds", type=float, default=12.0)
    a = ap.parse_args()
    make_dataset(a.out, a.seconds, log=print)
    print(write_config(os.path.join(a.out, "config"), a.out, os.path.join(a.out, "out", "traj.txt")))
