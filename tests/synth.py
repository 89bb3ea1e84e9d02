"""Seeded synthetic problems for the update half of the path (sizes follow SURVEY.md §8(d))."""
import numpy as np


def spd_cov(n, seed=7, lo=1e-6, hi=1e-2):
    """P = A A^T + 1e-3 I scaled so that diag lies in [lo, hi] (SURVEY §8(d) cfg B)."""
    rng = np.random.default_rng(seed)
    A = rng.normal(0, 1e-2, (n, n))
    P = A @ A.T + 1e-3 * np.eye(n) * 1e-2
    d = np.sqrt(np.diag(P))
    target = np.sqrt(np.exp(rng.uniform(np.log(lo), np.log(hi), n)))
    s = target / d
    P = P * s[:, None] * s[None, :]
    return 0.5 * (P + P.T)


def col_map(n, k, seed=3, skip=15):
    """k distinct state columns, skipping the first `skip` (IMU) entries when possible."""
    rng = np.random.default_rng(seed)
    pool = np.arange(skip, n) if n - skip >= k else np.arange(n)
    cols = np.sort(rng.choice(pool, size=k, replace=False)).astype(np.int32)
    return cols


def msckf_batch(F=70, M=15, k=98, fdim=3, seed=11, ragged=True, outlier_frac=0.15, ld=None, sigma=1.0):
    """Per-feature whitened systems (Hf [F,fdim,ld], Hx [F,k,ld], res [F,ld], rows [F]).
    Each feature touches a contiguous window of 6-wide clone blocks plus the last 8 columns
    (intrinsics), mimicking CamHelper::get_feature_jacobian_full's sparsity."""
    rng = np.random.default_rng(seed)
    ld = ld or 2 * M
    rows = np.full(F, 2 * M, dtype=np.int32)
    if ragged:
        rows = (2 * rng.integers(max(3, M // 3), M + 1, F)).astype(np.int32)
        rows[0] = 2 * M
    Hf = np.zeros((F, fdim, ld))
    Hx = np.zeros((F, k, ld))
    res = np.zeros((F, ld))
    nblk = max(1, (k - 8) // 6)
    for f in range(F):
        r = rows[f]
        Hf[f, :, :r] = rng.normal(0, 1.0, (fdim, r))
        nobs = r // 2
        start = rng.integers(0, max(1, nblk - nobs + 1))
        for o in range(nobs):
            b = min(nblk - 1, start + o)
            Hx[f, 6 * b:6 * b + 6, 2 * o:2 * o + 2] = rng.normal(0, 1.0, (6, 2))
            # interpolation spreads a little onto the neighbours
            if b + 1 < nblk:
                Hx[f, 6 * b + 6:6 * b + 12, 2 * o:2 * o + 2] = rng.normal(0, 0.05, (6, 2))
        if k >= 8:
            Hx[f, k - 8:, :r] = rng.normal(0, 0.3, (8, r))
        scale = 5.0 if rng.uniform() < outlier_frac else 0.35
        res[f, :r] = rng.normal(0, sigma * scale, r)
    return rows, Hf, Hx, res


def q95_table(n=1024):
    from scipy.stats import chi2
    t = np.zeros(n)
    t[1:] = chi2.ppf(0.95, np.arange(1, n))
    return t


# ------------------------------------------------------------------ images (SURVEY.md §8(d) cfg 2)
def texture_canvas(w, h, seed=42, margin=64, blobs=300, lines=0):
    """Band-limited noise texture (sigma_blur 1.5 px, contrast-stretched) + Harris-strong blobs,
    optionally straight high-contrast edges, on a canvas larger than the image by `margin`."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    W, H = w + 2 * margin, h + 2 * margin
    img = rng.normal(0, 1, (H, W))
    img = ndi.gaussian_filter(img, 1.5) * 3.0 + ndi.gaussian_filter(rng.normal(0, 1, (H, W)), 6.0) * 6.0
    for _ in range(blobs):
        x, y = rng.uniform(8, W - 8), rng.uniform(8, H - 8)
        s = rng.uniform(2.0, 4.0)
        amp = rng.choice([-1, 1]) * rng.uniform(1.0, 2.0)
        x0, x1, y0, y1 = int(x - s), int(x + s), int(y - s), int(y + s)
        img[y0:y1, x0:x1] += amp
    for _ in range(lines):
        L = rng.uniform(60, 300)
        th = rng.uniform(0, np.pi)
        cx, cy = rng.uniform(40, W - 40), rng.uniform(40, H - 40)
        n = int(L)
        t = np.linspace(-L / 2, L / 2, n)
        for off in range(-1, 2):
            xs = np.clip((cx + t * np.cos(th) - off * np.sin(th)).astype(int), 0, W - 1)
            ys = np.clip((cy + t * np.sin(th) + off * np.cos(th)).astype(int), 0, H - 1)
            img[ys, xs] += 3.0
    img = ndi.gaussian_filter(img, 0.8)
    lo, hi = np.percentile(img, [1, 99])
    img = np.clip((img - lo) / (hi - lo), 0, 1)
    return img


def render_frame(canvas, w, h, tx=0.0, ty=0.0, rot_deg=0.0, scale=1.0, margin=64):
    """Samples the canvas under a similarity warp about the image centre (cubic interpolation).
    Returns the u8 frame; a canvas point c maps to frame pixel p = s R (c - c0) + c0 + t."""
    from scipy import ndimage as ndi
    th = np.deg2rad(rot_deg)
    cx, cy = margin + w / 2.0, margin + h / 2.0
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    # inverse map: frame pixel -> canvas
    px, py = xs + margin - cx - tx, ys + margin - cy - ty
    c, s = np.cos(-th), np.sin(-th)
    qx = (c * px - s * py) / scale + cx
    qy = (s * px + c * py) / scale + cy
    out = ndi.map_coordinates(canvas, [qy, qx], order=3, mode="reflect")
    return np.clip(np.rint(out * 255.0), 0, 255).astype(np.uint8)


def warp_points(pts, w, h, tx, ty, rot_deg, scale):
    """Where frame-0 (identity warp) pixel positions land in the warped frame."""
    th = np.deg2rad(rot_deg)
    c0 = np.array([w / 2.0, h / 2.0])
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    return ((pts - c0) @ R.T) * scale + c0 + np.array([tx, ty])


def grid_points(w, h, n, seed=1, border=24):
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.uniform(border, w - border, n), rng.uniform(border, h - border, n)]).astype(np.float32)


EUROC_K8 = np.array([458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])


# ------------------------------------------------------------------ visual-inertial scene (SURVEY §8(d) cfg 2 filter problem)
def _exp_so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def radtan_distort(K8, xy):
    x, y = xy[..., 0], xy[..., 1]
    r2 = x * x + y * y
    rad = 1 + K8[4] * r2 + K8[5] * r2 * r2
    xd = x * rad + 2 * K8[6] * x * y + K8[7] * (r2 + 2 * x * x)
    yd = y * rad + K8[6] * (r2 + 2 * y * y) + 2 * K8[7] * x * y
    return np.stack([K8[0] * xd + K8[2], K8[1] * yd + K8[3]], axis=-1)


def vio_scene(n_clones=15, F=70, M=15, seed=3, noise_px=1.0, dt_clone=0.05, obs_offset=0.0, w=752, h=480,
              calib_int=True, fej_noise=0.0, depth_scale=1.0):
    """Smooth 1 m/s arc, clones every dt_clone seconds, F landmarks each observed in its last M (or fewer)
    clone frames at times clone_time + obs_offset.  State layout: IMU(15) | intrinsics(8) | clones(6 each)
    -> n = 113, k = 98 for 15 clones.  Returns a dict of plain numpy arrays."""
    rng = np.random.default_rng(seed)
    K8 = EUROC_K8.copy()
    t = 100.0 + dt_clone * np.arange(n_clones)
    R_ItoC = _exp_so3(np.array([0.01, -0.02, 0.015]))
    p_IinC = np.array([0.05, -0.02, 0.01])

    def pose(ti):
        s = ti - t[0]
        R = _exp_so3(np.array([0.03 * np.sin(1.3 * s), 0.08 * s, 0.02 * np.cos(0.7 * s)]))
        p = np.array([1.0 * s, 0.05 * np.sin(2.0 * s), 0.1 * s * s])
        return R, p

    poses = [pose(ti) for ti in t]
    clone_R = np.array([R for R, _ in poses])
    clone_p = np.array([p for _, p in poses])
    first = 15 + (8 if calib_int else 0)
    ids = (first + 6 * np.arange(n_clones)).astype(np.int32)
    n_state = first + 6 * n_clones
    # landmarks visible from every clone
    pts = []
    while len(pts) < F:
        c = depth_scale * np.array([rng.uniform(-20, 20), rng.uniform(-14, 14), rng.uniform(3, 60)])   # (depth_scale >> 1: weak parallax)
        ok = True
        for R, p in poses:
            pc = R_ItoC @ (R @ (c - p)) + p_IinC
            uv = radtan_distort(K8, pc[:2] / pc[2])
            ok = ok and pc[2] > 1 and 5 < uv[0] < w - 5 and 5 < uv[1] < h - 5 and np.hypot(*(pc[:2] / pc[2])) < 0.9
        if ok:
            pts.append(c)
    pts = np.array(pts)
    obs_ptr, obs_time, obs_uv, obs_clone = [0], [], [], []
    for f in range(F):
        m = M if f % 4 else max(3, M - rng.integers(0, M // 2 + 1))
        for ci in range(n_clones - m, n_clones):
            tm = t[ci] + (obs_offset if ci < n_clones - 1 else 0.0)
            R, p = (clone_R[ci], clone_p[ci]) if tm == t[ci] else pose(tm)
            pc = R_ItoC @ (R @ (pts[f] - p)) + p_IinC
            uv = radtan_distort(K8, pc[:2] / pc[2]) + rng.normal(0, noise_px, 2)
            obs_time.append(tm)
            obs_uv.append(uv)
            obs_clone.append(ci)
        obs_ptr.append(len(obs_time))
    clone_R_fej, clone_p_fej = clone_R.copy(), clone_p.copy()
    if fej_noise > 0:
        for i in range(n_clones):
            clone_R_fej[i] = _exp_so3(rng.normal(0, fej_noise, 3)) @ clone_R[i]
            clone_p_fej[i] = clone_p[i] + rng.normal(0, fej_noise, 3)
    return dict(t=t, R=clone_R, p=clone_p, Rf=clone_R_fej, pf=clone_p_fej, ids=ids, n_state=n_state, R_ItoC=R_ItoC,
                p_IinC=p_IinC, K8=K8, pts=pts, obs_ptr=np.array(obs_ptr, np.int32), obs_time=np.array(obs_time),
                obs_uv=np.array(obs_uv, np.float32), obs_clone=np.array(obs_clone), intr_id=15 if calib_int else -1,
                pose_fn=pose)


def scene_views(pkg, sc, p_FinG=None, sigma_pix=1.5, **kw):
    st = pkg.StateView(sc["t"], sc["R"], sc["p"], sc["ids"], sc["R_ItoC"], sc["p_IinC"], sc["K8"], clone_R_fej=sc["Rf"],
                       clone_p_fej=sc["pf"], intrinsic_state_id=sc["intr_id"], sigma_pix=sigma_pix, **kw)
    tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], sc["pts"] if p_FinG is None else p_FinG)
    return st, tr


def line_scene(sc, L=40, M=15, seed=5, noise_px=0.5, w=752, h=480, depth=(6.0, 40.0)):
    """3-D line segments seen from the clones of a vio_scene: per observation the raw pixel end points
    (pinhole projection: the reference's line model uses K only, LineHelper.cpp:861-864) and the
    normalised end points.  The visible end points slide along the line from view to view, as a
    detector's would.  Returns plain numpy arrays."""
    rng = np.random.default_rng(seed)
    K8, R_ItoC, p_IinC = sc["K8"], sc["R_ItoC"], sc["p_IinC"]
    n_clones = len(sc["t"])
    lines, obs_ptr, obs_time, seg_uv, seg_uvn = [], [0], [], [], []
    while len(lines) < L:
        a = np.array([rng.uniform(-15, 15), rng.uniform(-10, 10), rng.uniform(*depth)])
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        half = rng.uniform(1.0, 4.0)
        m = M if len(lines) % 3 else max(3, M - int(rng.integers(0, M // 2 + 1)))
        rows, ok = [], True
        for ci in range(n_clones - m, n_clones):
            R, p = sc["R"][ci], sc["p"][ci]
            ends = []
            for sgn in (-1.0, 1.0):
                P3 = a + sgn * half * (1 + 0.1 * rng.uniform(-1, 1)) * d
                pc = R_ItoC @ (R @ (P3 - p)) + p_IinC
                xn = pc[:2] / pc[2]
                uv = np.array([K8[0] * xn[0] + K8[2], K8[1] * xn[1] + K8[3]])
                ok = ok and pc[2] > 1 and 5 < uv[0] < w - 5 and 5 < uv[1] < h - 5
                ends.append((uv, xn))
            rows.append((sc["t"][ci], np.concatenate([ends[0][0], ends[1][0]]), np.concatenate([ends[0][1], ends[1][1]])))
        if not ok:
            continue
        for tm, uv, xn in rows:
            obs_time.append(tm)
            seg_uv.append(uv + rng.normal(0, noise_px, 4))
            seg_uvn.append(xn + rng.normal(0, noise_px / K8[0], 4))
        obs_ptr.append(len(obs_time))
        lines.append(np.concatenate([np.cross(a, d), d]))  # Pluecker: moment, direction
    return dict(lines=np.array(lines), obs_ptr=np.array(obs_ptr, np.int32), obs_time=np.array(obs_time),
                seg_uv=np.array(seg_uv, np.float32), seg_uvn=np.array(seg_uvn, np.float32))


def cpi_scene(sc, imu_dt=0.005, gravity=(0.0, 0.0, 9.81)):
    """State::cpis for the trajectory of a vio_scene: one record at every clone time (R = I, alpha = 0, the clone's
    velocity) and one every imu_dt after it until the next clone, all integrated from that clone.  The records hold
    the exact preintegrated quantities of the analytic trajectory: R_I0toIk = R_GtoIk R_GtoI0^T and
    alpha = R_GtoI0 (p_k - p_0 - v_0 dt + g dt^2 / 2), so get_interpolated_pose_imu reproduces pose_fn at them."""
    t, pose = sc["t"], sc["pose_fn"]
    g = np.asarray(gravity, dtype=np.float64)

    def vel(ti, h=1e-6):
        return (pose(ti + h)[1] - pose(ti - h)[1]) / (2 * h)

    rec = []
    for i, tc in enumerate(t):
        R0, p0 = sc["R"][i], sc["p"][i]
        v0 = vel(tc)
        rec.append((tc, tc, np.eye(3), np.zeros(3), v0))
        end = t[i + 1] if i + 1 < len(t) else tc + 4 * imu_dt
        k = 1
        while tc + k * imu_dt < end - 1e-9:
            tk = tc + k * imu_dt
            Rk, pk = pose(tk)
            dt = tk - tc
            rec.append((tk, tc, Rk @ R0.T, R0 @ (pk - p0 - v0 * dt + 0.5 * g * dt * dt), vel(tk)))
            k += 1
    return dict(t=np.array([r[0] for r in rec]), clone_t=np.array([r[1] for r in rec]), R=np.array([r[2] for r in rec]),
                alpha=np.array([r[3] for r in rec]), v=np.array([r[4] for r in rec]), gravity=g)


def imu_stream(pose_fn, t0, t1, rate=200.0, gravity=(0.0, 0.0, 9.81), bg=(0, 0, 0), ba=(0, 0, 0), h=1e-4):
    """Noise-free IMU samples of an analytic trajectory pose_fn(t) -> (R_GtoI, p_IinG), JPL conventions of the reference:
    R_GtoI(t + dt) = exp(-[w dt]x) R_GtoI(t),  a_m = R_GtoI (p'' + g)."""
    from scipy.spatial.transform import Rotation
    g = np.asarray(gravity, dtype=np.float64)
    n = int(round((t1 - t0) * rate)) + 1
    t = t0 + np.arange(n) / rate
    wm, am = np.zeros((n, 3)), np.zeros((n, 3))
    for i, ti in enumerate(t):
        Rm, _ = pose_fn(ti - h)
        Rp, _ = pose_fn(ti + h)
        R, _ = pose_fn(ti)
        wm[i] = -Rotation.from_matrix(Rp @ Rm.T).as_rotvec() / (2 * h) + np.asarray(bg)
        acc = (pose_fn(ti + h)[1] - 2 * pose_fn(ti)[1] + pose_fn(ti - h)[1]) / (h * h)
        am[i] = R @ (acc + g) + np.asarray(ba)
    return t, wm, am
