"""SURVEY §8(f) rank 3 on the CPU: the wheel-odometry restatement (3D types) against a simulated planar vehicle
(consistent clone poses give a zero residual), finite differences of its own residual, and the library's host-only
select_wheel_data against the oracle."""
import functools

import numpy as np
from scipy.spatial.transform import Rotation

import oracle_lib

RL, RR, B = 0.31, 0.305, 1.52
R_ITOO = Rotation.from_rotvec([0.02, -0.01, 0.6]).as_matrix()
P_IINO = np.array([0.4, -0.1, 0.25])


def vehicle(t):
    """yaw rate and forward speed of the odometry frame"""
    return 0.35 + 0.2 * np.sin(0.8 * t), 4.0 + 1.5 * np.cos(0.5 * t)


@functools.lru_cache(maxsize=None)
def odom_pose(t_end, t0=20.0, h=1e-4):
    """R_GtoO, p_OinG by fine RK4 of the unicycle from t0 (identity pose)"""
    def f(t, s):
        w, v = vehicle(t)
        return np.array([w, v * np.cos(s[0]), v * np.sin(s[0])])
    s, t = np.zeros(3), t0
    n = int(np.ceil((t_end - t0) / h))
    hh = (t_end - t0) / max(n, 1)
    for _ in range(n):
        k1 = f(t, s); k2 = f(t + hh / 2, s + hh / 2 * k1); k3 = f(t + hh / 2, s + hh / 2 * k2); k4 = f(t + hh, s + hh * k3)
        s = s + hh / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        t += hh
    R_OtoG = Rotation.from_rotvec([0, 0, s[0]]).as_matrix()
    return R_OtoG.T, np.array([s[1], s[2], 0.0])


def imu_pose(t):
    R_GtoO, p_O = odom_pose(t)
    R_GtoI = R_ITOO.T @ R_GtoO
    p_OinI = -R_ITOO.T @ P_IINO
    return R_GtoI, p_O - R_GtoI.T @ p_OinI


def wheel_stream(t0, t1, rate=100.0, kind=0):
    n = int(round((t1 - t0) * rate)) + 1
    t = t0 + np.arange(n) / rate
    wv = np.array([vehicle(x) for x in t])
    w, v = wv[:, 0], wv[:, 1]
    if kind % 3 == 0:    # Wheel3DAng / Wheel2DAng: wheel angular velocities
        return t, (v - w * B / 2) / RL, (v + w * B / 2) / RR
    if kind % 3 == 1:    # ...Lin: wheel linear velocities
        return t, v - w * B / 2, v + w * B / 2
    return t, w, v       # ...Cen


def make(pkg, kind=0, ext=False, dt=False, intr=False, t0=20.3, t1=20.8, d0=None, d1=None, intr_v=(RL, RR, B)):
    opt = pkg.PlvWheelOptions(kind, 0.2, 0.5, 0.1, int(ext), int(dt), int(intr), 2.0)
    R0, p0 = imu_pose(t0)
    R1, p1 = imu_pose(t1)
    if d0 is not None:
        R0, p0 = Rotation.from_rotvec(-d0[:3]).as_matrix() @ R0, p0 + d0[3:]
    if d1 is not None:
        R1, p1 = Rotation.from_rotvec(-d1[:3]).as_matrix() @ R1, p1 + d1[3:]
    st = pkg.PlvWheelState.make(intr_v, R_ITOO, P_IINO, R0, p0, R1, p1, 15, 27, w0=(0.01, 0.02, 0.4), v0=(3.0, 2.0, 0.1), w1=(0.0, 0.01, 0.5),
                                v1=(2.0, 3.5, 0.0), ext_id=3 if ext else -1, dt_id=9 if dt else -1, intr_id=10 if intr else -1)
    return opt, st


def test_select_wheel_data(pkg):
    po = oracle_lib.load_prop(pkg)
    rng = np.random.default_rng(0)
    t = 20.0 + 0.01 * np.arange(200) + rng.uniform(-0.002, 0.002, 200)
    m1, m2 = rng.normal(size=200), rng.normal(size=200)
    for a, b in ((20.3007, 20.8004), (t[30], t[80]), (20.3007, 20.3052), (t[30] + 1e-13, t[31])):
        ok, ot, o1, o2 = po.select_wheel_data(t, m1, m2, a, b)
        ok2, ot2, o12, o22 = pkg.select_wheel_data(t, m1, m2, a, b)
        assert ok and ok2 and np.array_equal(ot, ot2) and np.array_equal(o1, o12) and np.array_equal(o2, o22)
        assert ot[-1] == b and abs(ot[0] - a) < 1e-12 and (np.diff(ot) >= 1e-12).all()
        assert set(t[(t > a) & (t < b)]) <= set(ot)
    for a, b in ((19.0, 20.5), (20.5, 22.5), (20.5, t[-1])):
        assert not po.select_wheel_data(t, m1, m2, a, b)[0] and not pkg.select_wheel_data(t, m1, m2, a, b)[0]
    assert not pkg.select_wheel_data(t[:0], m1[:0], m2[:0], 20.1, 20.2)[0]


def test_consistent_poses_give_zero_residual(pkg):
    po = oracle_lib.load_prop(pkg)
    for kind in (0, 1, 2):
        t, m1, m2 = wheel_stream(20.0, 21.0, kind=kind)
        opt, st = make(pkg, kind)
        ok, st_, s1, s2 = po.select_wheel_data(t, m1, m2, 20.3, 20.8)
        assert ok
        H, res, Cov, cols, R3, p3 = po.wheel_linear_system(opt, st, st_, s1, s2)
        assert H.shape == (6, 12) and list(cols) == list(range(15, 21)) + list(range(27, 33))
        assert np.abs(res).max() < 5e-6          # 2 m travelled; 100 Hz piecewise-linear wheel speeds, RK4
        R_O0, p_O0 = odom_pose(20.3)
        R_O1, p_O1 = odom_pose(20.8)
        assert np.abs(R3 - R_O1 @ R_O0.T).max() < 1e-6 and np.abs(p3 - R_O0 @ (p_O1 - p_O0)).max() < 5e-6
        assert np.abs(Cov - Cov.T).max() == 0 and np.linalg.eigvalsh(Cov).min() > 0
        # white noise integrates to sigma^2 * T; the vehicle only yaws, so the zz entry and the xy trace are invariant
        sig2 = {0: 0.2 ** 2, 1: 0.5 ** 2 / B ** 2, 2: 0.2 ** 2}[kind]
        assert abs(Cov[2, 2] / (0.1 ** 2 * 0.5) - 1) < 1e-9 and abs((Cov[0, 0] + Cov[1, 1]) / ((sig2 + 0.1 ** 2) * 0.5) - 1) < 1e-9


def test_jacobians_by_finite_differences(pkg):
    po = oracle_lib.load_prop(pkg)
    t, m1, m2 = wheel_stream(20.0, 21.0)
    _, st_, s1, s2 = po.select_wheel_data(t, m1, m2, 20.3, 20.8)
    opt, st = make(pkg, 0, intr=True)
    H, res0, _, cols, _, _ = po.wheel_linear_system(opt, st, st_, s1, s2)
    assert H.shape == (6, 15) and list(cols[12:]) == [10, 11, 12]
    eps = 1e-6
    J = np.zeros((6, 15))
    for k in range(15):
        d = np.zeros(15)
        d[k] = eps
        r = []
        for sgn in (1, -1):
            dd = sgn * d
            intr_v = np.array([RL, RR, B]) + dd[12:]
            o2, s2_ = make(pkg, 0, intr=True, d0=dd[:6], d1=dd[6:12], intr_v=intr_v)
            # the measurement is preintegrated from wheel readings: readings stay, the intrinsics move
            r.append(po.wheel_linear_system(o2, s2_, st_, s1, s2)[1])
        J[:, k] = -(r[0] - r[1]) / (2 * eps)   # res = z - h(x):  dres/dx = -H
    assert np.abs(J[:, :12] - H[:, :12]).max() < 1e-6
    # the intrinsic columns come from a first-order recursion on the sample at the start of each step
    # (preintegration_intrinsics_3D), the measurement itself from RK4 over both ends: a few percent apart
    assert np.abs(J[:, 12:] - H[:, 12:]).max() < 0.05 * np.abs(H[:, 12:]).max()


def test_calibration_blocks_and_types(pkg):
    po = oracle_lib.load_prop(pkg)
    t, m1, m2 = wheel_stream(20.0, 21.0)
    _, st_, s1, s2 = po.select_wheel_data(t, m1, m2, 20.3, 20.8)
    opt, st = make(pkg, 0, ext=True, dt=True, intr=True)
    H, res, Cov, cols, _, _ = po.wheel_linear_system(opt, st, st_, s1, s2)
    assert H.shape == (6, 22) and list(cols[12:]) == [3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
    # the time-offset column is the pose Jacobians applied to the clone velocities (UpdaterWheel.cpp:404-413)
    w0, v0, w1, v1 = np.array(st.w0), np.array(st.v0), np.array(st.w1), np.array(st.v1)
    assert np.abs(H[:, 18] - (H[:, 0:3] @ w0 + H[:, 3:6] @ v0 + H[:, 6:9] @ w1 + H[:, 9:12] @ v1)).max() < 1e-12
    # extrinsic block: dzr/dth = I - R_O0toO1, dzp/dp = I - R_O1toO0
    R_O0toO1 = R_ITOO @ imu_pose(20.8)[0] @ imu_pose(20.3)[0].T @ R_ITOO.T
    assert np.abs(H[0:3, 12:15] - (np.eye(3) - R_O0toO1)).max() < 1e-12 and np.abs(H[3:6, 15:18] - (np.eye(3) - R_O0toO1.T)).max() < 1e-12


def test_2d_types(pkg):
    """Wheel2DAng / Lin / Cen: yaw + planar translation (3 rows).  Level odometry frame, so the planar model is exact."""
    po = oracle_lib.load_prop(pkg)
    global R_ITOO
    keep = R_ITOO
    try:
        R_ITOO = Rotation.from_rotvec([0.0, 0.0, 0.6]).as_matrix()   # the 2D model assumes the wheel frame's z is the yaw axis
        odom_pose.cache_clear()
        for kind in (3, 4, 5):
            t, m1, m2 = wheel_stream(20.0, 21.0, rate=400.0, kind=kind)
            ok, st_, s1, s2 = po.select_wheel_data(t, m1, m2, 20.3, 20.8)
            opt, st = make(pkg, kind, ext=True, dt=True, intr=(kind == 3))
            H, res, Cov, cols, _, meas = po.wheel_linear_system(opt, st, st_, s1, s2)
            kk = 19 + (3 if kind == 3 else 0)
            assert H.shape == (3, kk) and Cov.shape == (3, 3) and len(res) == 3
            # an independent transcription of preintegration_2D's mean recursion (UpdaterWheel.cpp:502-571)
            th = x = y = 0.0
            for i in range(len(st_) - 1):
                dt = st_[i + 1] - st_[i]
                if kind == 3:
                    wv = [((s2[j] * RR - s1[j] * RL) / B, (s2[j] * RR + s1[j] * RL) / 2) for j in (i, i + 1)]
                elif kind == 4:
                    wv = [((s2[j] - s1[j]) / B, (s2[j] + s1[j]) / 2) for j in (i, i + 1)]
                else:
                    wv = [(s1[j], s2[j]) for j in (i, i + 1)]
                (w1, v1), (w2, v2) = wv
                wa, vj = (w2 - w1) / dt, (v2 - v1) / dt
                w, v = w1, v1
                k1t, k1x = -w * dt, v * dt
                w, v = w + 0.5 * wa * dt, v + 0.5 * vj * dt
                k2t, k2x = -w * dt, v * np.cos(0.5 * k1t) * dt
                k3t, k3x = -w * dt, v * np.cos(0.5 * k2t) * dt
                w, v = w + 0.5 * wa * dt, v + 0.5 * vj * dt
                k4t, k4x = -w * dt, v * np.cos(k3t) * dt
                y = y - v1 * np.sin(th - w1 * dt) * dt if abs(w1) < 1e-4 else y - (v1 * (np.cos(th - w1 * dt) - np.cos(th))) / w1
                x = x + (k1x + 2 * k2x + 2 * k3x + k4x) / 6
                th = th + (k1t + 2 * k2t + 2 * k3t + k4t) / 6
            assert np.abs(meas - [th, x, y]).max() < 1e-12
            # against the vehicle's true motion: the yaw is exact; the reference integrates x in the frame of each step's start and
            # y with a first-order closed form, so the translation is only good to centimetres over half a second of turning
            R_O0, p_O0 = odom_pose(20.3)
            R_O1, p_O1 = odom_pose(20.8)
            d = R_O0 @ (p_O1 - p_O0)
            yaw = Rotation.from_matrix(R_O1 @ R_O0.T).as_rotvec()[2]
            assert abs(meas[0] - yaw) < 1e-6 and abs(meas[1] - d[0]) < 0.05 and abs(meas[2] - d[1]) < 0.05
            assert abs(res[0]) < 1e-6 and abs(res[1] - (meas[1] - d[0])) < 1e-6 and abs(res[2] - (meas[2] - d[1])) < 1e-6
            assert np.abs(Cov - Cov.T).max() == 0 and np.linalg.eigvalsh(Cov).min() > 0
            # pose Jacobians by finite differences of the residual
            eps = 1e-6
            J = np.zeros((3, 12))
            for c in range(12):
                dd = np.zeros(12)
                dd[c] = eps
                r = []
                for sgn in (1, -1):
                    o2, s2_ = make(pkg, kind, ext=True, dt=True, intr=(kind == 3), d0=sgn * dd[:6], d1=sgn * dd[6:])
                    r.append(po.wheel_linear_system(o2, s2_, st_, s1, s2)[1])
                J[:, c] = -(r[0] - r[1]) / (2 * eps)
            assert np.abs(J - H[:, :12]).max() < 1e-5
            # the time-offset column = pose Jacobians applied to the clone velocities
            w0, v0, w1, v1 = np.array(st.w0), np.array(st.v0), np.array(st.w1), np.array(st.v1)
            assert np.abs(H[:, 18] - (H[:, 0:3] @ w0 + H[:, 3:6] @ v0 + H[:, 6:9] @ w1 + H[:, 9:12] @ v1)).max() < 1e-12
    finally:
        R_ITOO = keep
        odom_pose.cache_clear()
