"""SURVEY §8(f) rank 4, initialisers: the library's host code (plv_init_imu_static, plv_init_imu_wheel) against the numpy
restatement in oracle/init_oracle.py on simulated data, and both against the simulated truth.  No device work: runs on the CPU."""
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import init_oracle as io  # noqa: E402

G = np.array([0.0, 0.0, 9.81])
RL, RR, B = 0.31, 0.305, 1.52
R_ITOO = Rotation.from_rotvec([0.05, -0.08, 0.3]).as_matrix()   # a tilted IMU: gravity is not along its z axis
P_IINO = np.array([0.4, -0.1, 0.25])
BG, BA = np.array([0.004, -0.003, 0.002]), np.array([0.05, -0.03, 0.04])
RADIUS = 8.0


def arc(t, moving):
    """arc length, speed of a vehicle on a circle: smooth start from rest when `moving`"""
    if not moving:
        return 0.0, 0.0
    return 3.0 * t + 1.2 * np.sin(0.9 * t) / 0.9, 3.0 + 1.2 * np.cos(0.9 * t)


def make_pose_fn(moving):
    def odom(t):
        s, _ = arc(t, moving)
        th = s / RADIUS
        R_OtoG = Rotation.from_rotvec([0, 0, th]).as_matrix()
        if moving:
            # body roll / pitch on the suspension: without any rotation out of the road plane the accelerometer bias and gravity
            # along the yaw axis cannot be separated (the constrained solve of IW_Initializer.cpp:266-432 becomes singular)
            R_OtoG = R_OtoG @ Rotation.from_rotvec([0, 0.003 * np.sin(2.1 * t), 0]).as_matrix() @ Rotation.from_rotvec([0.002 * np.sin(1.7 * t + 0.5), 0, 0]).as_matrix()
        return R_OtoG.T, np.array([RADIUS * np.sin(th), RADIUS * (1 - np.cos(th)), 0.0])

    def imu(t):
        R_GtoO, p_O = odom(t)
        R_GtoI = R_ITOO.T @ R_GtoO
        return R_GtoI, p_O + R_GtoI.T @ (R_ITOO.T @ P_IINO)   # p_I = p_O - R_ItoG p_OinI, p_OinI = -R_OtoI p_IinO
    return odom, imu


def streams(moving, t1=6.0, seed=0, gyro_noise=0.0):
    odom, imu = make_pose_fn(moving)
    t, wm, am = synth.imu_stream(imu, 0.0, t1, rate=200.0, bg=BG, ba=BA)
    rng = np.random.default_rng(seed)
    wm = wm + rng.normal(0, gyro_noise, wm.shape)
    tw = 0.003 + np.arange(int(t1 * 50)) / 50.0
    wv = np.array([[arc(x, moving)[1] / RADIUS, arc(x, moving)[1]] for x in tw])
    m1, m2 = (wv[:, 1] - wv[:, 0] * B / 2) / RL, (wv[:, 1] + wv[:, 0] * B / 2) / RR
    return (t, wm, am), (tw, m1, m2), imu


def run_both(pkg, imu, whl, aligned, start=1.0, step=0.25, threshold=0.1, gravity=G):
    """Feeds growing buffers to both implementations, as SystemManager::feed_measurement_imu does while not initialised."""
    t, wm, am = imu
    tw, m1, m2 = whl
    hip = pkg.IwInitializer("Wheel3DAng", (RL, RR, B), R_ITOO, P_IINO, 0.0, threshold, gravity, aligned)
    orc = io.IWInitializer("Wheel3DAng", (RL, RR, B), R_ITOO, P_IINO, 0.0, threshold, gravity, aligned)
    calls = 0
    tk = start
    while tk <= t[-1]:
        ni, nw = int(np.searchsorted(t, tk)), int(np.searchsorted(tw, tk))
        lo_i, lo_w = max(0, ni - 400), max(0, nw - 100)     # a 2 s window, as delete_old_measurements keeps it bounded
        a = hip.initialization(t[lo_i:ni], wm[lo_i:ni], am[lo_i:ni], tw[lo_w:nw], m1[lo_w:nw], m2[lo_w:nw])
        b = orc.initialization(t[lo_i:ni], wm[lo_i:ni], am[lo_i:ni], tw[lo_w:nw], m1[lo_w:nw], m2[lo_w:nw])
        calls += 1
        assert hip.last_mode == orc.last_mode
        assert (hip.last_init is None) == (orc.last_init is None), tk
        if orc.last_init is not None:
            assert np.abs(hip.last_init - orc.last_init).max() < 1e-7, (tk, hip.last_init - orc.last_init)
        assert hip.state.cnt_smooth == orc.cnt_smooth, tk
        assert (a is None) == (b is None), tk
        if a is not None:
            assert np.abs(a - b).max() < 1e-7
            return a, calls, hip, orc
        tk += step
    return None, calls, hip, orc


def check_against_truth(x, imu_fn, tol_ba, tol_g):
    t0 = x[0]
    R_GtoI, _ = imu_fn(t0)
    v_true = (imu_fn(t0 + 1e-5)[1] - imu_fn(t0 - 1e-5)[1]) / 2e-5
    R_est = io.quat_2_Rot(x[1:5])
    # gravity direction in the IMU frame: third column of R_GtoI0 against the true one (yaw is free)
    assert np.linalg.norm(R_est[:, 2] - R_GtoI[:, 2]) < tol_g
    assert np.abs(x[5:8]).max() == 0
    assert np.abs(R_est @ x[8:11] - R_GtoI @ v_true).max() < 0.05      # velocity in the body frame
    assert np.abs(x[11:14] - BG).max() < 2e-3
    assert np.abs(x[14:17] - BA).max() < tol_ba


def test_imu_wheel_dynamic(pkg):
    imu, whl, imu_fn = streams(moving=True)
    x, calls, hip, orc = run_both(pkg, imu, whl, aligned=False)
    assert x is not None and calls >= 5 and hip.last_mode == 1
    check_against_truth(x, imu_fn, tol_ba=0.06, tol_g=1e-2)


def test_imu_wheel_static(pkg):
    imu, whl, imu_fn = streams(moving=False)
    x, calls, hip, orc = run_both(pkg, imu, whl, aligned=False)
    assert x is not None and hip.last_mode == 0
    # standing still, accelerometer bias and gravity cannot be told apart: the static path takes the whole specific force as
    # gravity (REF: IW_Initializer.cpp:52-58 comment) and the bias absorbs the difference in norm
    R_GtoI, _ = imu_fn(x[0])
    f = R_GtoI @ G + BA
    R_est = io.quat_2_Rot(x[1:5])
    assert np.linalg.norm(R_est[:, 2] - f / np.linalg.norm(f)) < 1e-6
    assert np.abs(x[11:14] - BG).max() < 1e-6 and np.abs(x[8:11]).max() < 1e-9


def test_imu_wheel_gravity_aligned_flag(pkg):
    """init.imu_gravity_aligned (the shipped KAIST configuration): gravity in {I0} is the configured vector."""
    imu, whl, _ = streams(moving=True)
    x, _, hip, _ = run_both(pkg, imu, whl, aligned=True, threshold=0.5)
    assert x is not None
    assert np.abs(hip.last_init[6:9] - G).max() == 0
    assert np.abs(io.quat_2_Rot(x[1:5]) - np.eye(3)).max() < 1e-12


def test_imu_wheel_waits_for_data_and_smoothness(pkg):
    imu, whl, _ = streams(moving=True, gyro_noise=0.0)
    t, wm, am = imu
    tw, m1, m2 = whl
    hip = pkg.IwInitializer("Wheel3DAng", (RL, RR, B), R_ITOO, P_IINO, 0.0, 0.1, G, False)
    assert hip.initialization(t[:2], wm[:2], am[:2], tw[:2], m1[:2], m2[:2]) is None and hip.last_mode == -1
    assert hip.initialization(t[:30], wm[:30], am[:30], tw[:8], m1[:8], m2[:8]) is None and hip.last_mode == -1   # < 20 wheel readings
    assert hip.state.cnt_smooth == -1
    # a threshold nothing can meet: every attempt resets the smoothness count
    hip0 = pkg.IwInitializer("Wheel3DAng", (RL, RR, B), R_ITOO, P_IINO, 0.0, 1e-9, G, False)
    for k in range(6):
        n, m = 400 + 40 * k, 100 + 10 * k
        assert hip0.initialization(t[:n], wm[:n], am[:n], tw[:m], m1[:m], m2[:m]) is None
    assert hip0.state.cnt_smooth == 0


def test_imu_wheel_other_types(pkg):
    imu, (tw, m1, m2), _ = streams(moving=True, t1=4.0)
    t, wm, am = imu
    w = (m2 * RR - m1 * RL) / B
    v = (m2 * RR + m1 * RL) / 2
    for typ, a, b in (("Wheel3DLin", v - w * B / 2, v + w * B / 2), ("Wheel2DCen", w, v)):
        hip = pkg.IwInitializer(typ, (RL, RR, B), R_ITOO, P_IINO, 0.0, 0.1, G, False)
        orc = io.IWInitializer(typ, (RL, RR, B), R_ITOO, P_IINO, 0.0, 0.1, G, False)
        hip.initialization(t[:500], wm[:500], am[:500], tw[:120], a[:120], b[:120])
        orc.initialization(t[:500], wm[:500], am[:500], tw[:120], a[:120], b[:120])
        assert orc.last_init is not None and np.abs(hip.last_init - orc.last_init).max() < 1e-7


def test_imu_static_init(pkg):
    """Stationary for 3 s, then a jerk: I_Initializer waits for the jerk and initialises from the still window before it."""
    rng = np.random.default_rng(3)
    R_GtoI = Rotation.from_rotvec([0.2, -0.1, 0.4]).as_matrix()
    t = np.arange(0, 5.0, 0.005)
    am = np.tile(R_GtoI @ G + BA, (len(t), 1)) + rng.normal(0, 0.01, (len(t), 3))
    wm = np.tile(BG, (len(t), 1)) + rng.normal(0, 0.001, (len(t), 3))
    moving = t > 3.0
    am[moving] += np.column_stack([2.0 * np.sin(7 * t[moving]), 1.0 * np.cos(5 * t[moving]), 0.5 * np.sin(3 * t[moving])])
    got = []
    for tk in np.arange(0.5, 5.0, 0.25):
        n = int(np.searchsorted(t, tk))
        a = pkg.init_imu_static(t[:n], wm[:n], am[:n], 1.0, 0.3, G)
        b = io.imu_static_init(t[:n], wm[:n], am[:n], 1.0, 0.3, G)
        assert (a is None) == (b is None), tk
        if a is not None:
            assert np.abs(a - b).max() < 1e-12
            got.append((tk, a))
    assert got and got[0][0] > 3.0 and got[0][0] <= 4.25
    x = got[0][1]
    R_est = io.quat_2_Rot(x[1:5])
    f = R_GtoI @ G + BA
    assert np.linalg.norm(R_est[:, 2] - f / np.linalg.norm(f)) < 2e-3
    assert np.abs(x[11:14] - BG).max() < 5e-4 and np.abs(x[5:11]).max() == 0
    assert np.abs((x[14:17] + R_est @ G) - f).max() < 5e-3       # ba = mean specific force - R g
    # never with a buffer shorter than two windows, never without the jerk
    assert pkg.init_imu_static(t[:300], wm[:300], am[:300], 1.0, 0.3, G) is None
    assert pkg.init_imu_static(t[:580], wm[:580], am[:580], 1.0, 0.3, G) is None


def test_constraint_polynomial_roots(pkg):
    """The norm-constrained gravity solve on its own: whatever the data, the returned gravity has the configured norm."""
    imu, whl, _ = streams(moving=True, gyro_noise=2e-3, seed=4)
    t, wm, am = imu
    tw, m1, m2 = whl
    hip = pkg.IwInitializer("Wheel3DAng", (RL, RR, B), R_ITOO, P_IINO, 0.0, 0.5, G, False)
    n_ok = 0
    for k in range(8):
        n, m = 420 + 60 * k, 105 + 15 * k
        hip.initialization(t[n - 400:n], wm[n - 400:n], am[n - 400:n], tw[m - 100:m], m1[m - 100:m], m2[m - 100:m])
        if hip.last_init is not None:
            n_ok += 1
            assert abs(np.linalg.norm(hip.last_init[6:9]) - 9.81) < 1e-3
    assert n_ok >= 6


def test_argument_checks(pkg):
    import ctypes as C
    lib = pkg.load_library()
    z = np.zeros(3)
    out, ok = np.zeros(17), C.c_int(7)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.plv_init_imu_static(5, None, None, None, 1.0, 0.3, dp(z), dp(out), C.byref(ok)) == pkg.PLV_E_BADARG
    assert lib.plv_init_imu_static(0, None, None, None, 1.0, 0.3, dp(z), dp(out), C.byref(ok)) == pkg.PLV_OK and ok.value == 0
    assert lib.plv_init_imu_static(0, None, None, None, 1.0, 0.3, dp(z), dp(out), None) == pkg.PLV_E_BADARG
    init = pkg.IwInitializer("Wheel3DAng", (RL, RR, B), R_ITOO, P_IINO, 0.0, 0.1, G, False)
    init.opt.wheel_type = 9
    with pytest.raises(pkg.PlvError):
        init.initialization(np.zeros(4), np.zeros((4, 3)), np.zeros((4, 3)), np.zeros(4), np.zeros(4), np.zeros(4))
    with pytest.raises(KeyError):
        pkg.IwInitializer("Wheel5D", (RL, RR, B), R_ITOO, P_IINO, 0.0, 0.1, G, False)
    # a fresh state starts at cnt_smooth = -1 (IW_Initializer.h) and a reset brings it back
    init2 = pkg.IwInitializer("Wheel2DCen", (RL, RR, B), R_ITOO, P_IINO, 0.0, 0.1, G, True)
    assert init2.state.cnt_smooth == -1
    init2.state.cnt_smooth = 3
    lib.plv_iw_init_reset(C.byref(init2.state))
    assert init2.state.cnt_smooth == -1 and not any(init2.state.prev_init)
