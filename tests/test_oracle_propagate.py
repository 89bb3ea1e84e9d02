"""SURVEY §8(f) rank 2 on the CPU: the propagation oracle against an analytic trajectory, finite differences of its own
mean propagation, a numpy EKFPropagation, and the library's host-only entry points against the oracle."""
import ctypes as C
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

import oracle_lib
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import eval_oracle as eo  # noqa: E402

G = np.array([0.0, 0.0, 9.81])


def traj(t):
    s = t - 50.0
    R = Rotation.from_rotvec([0.2 * np.sin(0.9 * s), 0.15 * np.cos(0.6 * s), 0.5 * s]).as_matrix().T   # R_GtoI
    p = np.array([2.0 * s + 0.3 * np.sin(1.1 * s), 0.5 * np.sin(0.7 * s), 0.2 * np.cos(0.5 * s)])
    return R, p


def vel(t, h=1e-5):
    return (traj(t + h)[1] - traj(t - h)[1]) / (2 * h)


def imu_at(pkg, t, bg=(0, 0, 0), ba=(0, 0, 0)):
    R, p = traj(t)
    return pkg.PlvImuState.make(eo.rot_2_quat(R), p, vel(t), bg, ba)


def test_select_imu_readings(pkg):
    po = oracle_lib.load_prop(pkg)
    rng = np.random.default_rng(0)
    t = 10.0 + 0.005 * np.arange(100) + rng.uniform(-0.001, 0.001, 100)
    wm, am = rng.normal(size=(100, 3)), rng.normal(size=(100, 3))
    for t0, t1 in ((10.1003, 10.2507), (t[20], t[40]), (t[20], t[21]), (10.1003, 10.1004), (t[0], t[-1])):
        ok, ot, ow, oa = po.select_imu_readings(t, wm, am, t0, t1)
        ok2, ot2, ow2, oa2 = pkg.select_imu_readings(t, wm, am, t0, t1)
        assert ok and ok2 and np.array_equal(ot, ot2) and np.array_equal(ow, ow2) and np.array_equal(oa, oa2)
        assert ot[0] == t0 and ot[-1] == t1 and (np.diff(ot) >= 0).all()
        # the middle loop keeps sample i only when sample i + 1 is still before time1 (Propagator.cpp:136): the last
        # buffer sample before time1 is skipped
        inner = [t[i] for i in range(len(t) - 1) if t0 < t[i] and t[i + 1] < t1]
        assert list(ot[1:-1]) == inner
        i = np.searchsorted(t, t0, side="right") - 1
        lam = (t0 - t[i]) / (t[i + 1] - t[i])
        assert np.allclose(ow[0], (1 - lam) * wm[i] + lam * wm[i + 1], atol=1e-13)
    for t0, t1 in ((10.2, 10.1), (10.2, 10.2), (9.0, 10.2), (10.2, 11.0)):
        assert not po.select_imu_readings(t, wm, am, t0, t1)[0]
        assert not pkg.select_imu_readings(t, wm, am, t0, t1)[0]
    assert not pkg.select_imu_readings(t[:1], wm[:1], am[:1], t[0], t[0] + 1e-3)[0]


def test_mean_propagation_follows_the_trajectory(pkg):
    po = oracle_lib.load_prop(pkg)
    bg, ba = (0.01, -0.02, 0.005), (0.05, 0.02, -0.03)
    t, wm, am = synth.imu_stream(traj, 50.0, 51.0, rate=200.0, bg=bg, ba=ba)
    imu = imu_at(pkg, 50.0, bg, ba)
    Phi, Qd, _, _ = po.propagate(imu, pkg.imu_noise(), t, wm, am)
    R1, p1 = traj(51.0)
    assert np.abs(eo.quat_2_rot(np.array(imu.q)) - R1).max() < 1e-6
    # 2.3 m travelled in 1 s; piecewise-linear IMU between 200 Hz samples + finite-difference samples: micrometres
    assert np.abs(np.array(imu.p) - p1).max() < 1e-5 and np.abs(np.array(imu.v) - vel(51.0)).max() < 1e-5
    assert list(imu.q) == list(imu.q_fej) and list(imu.p) == list(imu.p_fej)   # set_fej(imu_x), Propagator.cpp:237
    assert np.abs(Qd - Qd.T).max() == 0 and np.linalg.eigvalsh(Qd).min() > -1e-18
    # the chain in two halves composes: Phi = Phi_2 Phi_1, Qd = Phi_2 Qd_1 Phi_2^T + Qd_2
    imu2 = imu_at(pkg, 50.0, bg, ba)
    Pa, Qa, _, _ = po.propagate(imu2, pkg.imu_noise(), t[:101], wm[:101], am[:101])
    Pb, Qb, _, _ = po.propagate(imu2, pkg.imu_noise(), t[100:], wm[100:], am[100:])
    assert np.abs(Pb @ Pa - Phi).max() < 1e-12 * np.abs(Phi).max() and np.abs(Pb @ Qa @ Pb.T + Qb - Qd).max() < 1e-16
    assert np.abs(imu2.vec() - imu.vec()).max() < 1e-12


def _perturbed(pkg, imu0, dx):
    """x_true = x_est [+] dx in the reference's error state: R_true = exp(-[dth]x) R_est, the rest additive."""
    R = Rotation.from_rotvec(-dx[0:3]).as_matrix() @ eo.quat_2_rot(np.array(imu0.q))
    return pkg.PlvImuState.make(eo.rot_2_quat(R), np.array(imu0.p) + dx[3:6], np.array(imu0.v) + dx[6:9], np.array(imu0.bg) + dx[9:12],
                                np.array(imu0.ba) + dx[12:15])


def _error(imu_true, imu_est):
    dR = eo.quat_2_rot(np.array(imu_true.q)) @ eo.quat_2_rot(np.array(imu_est.q)).T
    return np.concatenate([-Rotation.from_matrix(dR).as_rotvec(), np.array(imu_true.p) - np.array(imu_est.p),
                           np.array(imu_true.v) - np.array(imu_est.v), np.array(imu_true.bg) - np.array(imu_est.bg),
                           np.array(imu_true.ba) - np.array(imu_est.ba)])


def test_phi_is_the_error_state_jacobian(pkg):
    po = oracle_lib.load_prop(pkg)
    t, wm, am = synth.imu_stream(traj, 50.0, 50.1, rate=200.0)
    nz = pkg.imu_noise()
    nom = imu_at(pkg, 50.0, (0.01, 0, 0), (0, 0.02, 0))
    base = nom.copy()
    Phi, _, _, _ = po.propagate(nom, nz, t, wm, am)
    J = np.zeros((15, 15))
    eps = 1e-6
    for k in range(15):
        dx = np.zeros(15)
        dx[k] = eps
        a, b = _perturbed(pkg, base, dx), _perturbed(pkg, base, -dx)
        po.propagate(a, nz, t, wm, am)
        po.propagate(b, nz, t, wm, am)
        J[:, k] = (_error(a, nom) - _error(b, nom)) / (2 * eps)
    # pose / velocity blocks: the analytic Phi IS the Jacobian of the RK4 mean (1e-9); the bias columns are the reference's
    # first-order model (no v-bg / p-bg coupling inside one IMU step), good to a few percent of the block
    assert np.abs(J[:9, :9] - Phi[:9, :9]).max() < 1e-7
    assert np.abs(J[:3, 9:12] - Phi[:3, 9:12]).max() < 1e-6
    for rb, cb in ((3, 9), (6, 9), (3, 12), (6, 12)):
        blk = np.abs(Phi[rb:rb + 3, cb:cb + 3]).max()
        assert np.abs(J[rb:rb + 3, cb:cb + 3] - Phi[rb:rb + 3, cb:cb + 3]).max() < 0.1 * blk
    assert np.abs(Phi[9:, 9:] - np.eye(6)).max() == 0 and np.abs(Phi[9:, :9]).max() == 0


def test_ekf_propagation_and_clone(pkg):
    po = oracle_lib.load_prop(pkg)
    t, wm, am = synth.imu_stream(traj, 50.0, 50.05, rate=200.0)
    n = 63
    P = synth.spd_cov(n, seed=3) * 1e-3
    for imu_id in (0, 12):
        imu = imu_at(pkg, 50.0)
        Phi, Qd, _, P1 = po.propagate(imu, pkg.imu_noise(), t, wm, am, P=P, imu_id=imu_id)
        Pf = np.eye(n)
        Pf[imu_id:imu_id + 15, imu_id:imu_id + 15] = Phi
        Qf = np.zeros((n, n))
        Qf[imu_id:imu_id + 15, imu_id:imu_id + 15] = Qd
        ref = Pf @ P @ Pf.T + Qf
        assert np.abs(P1 - ref).max() < 1e-15 * n and np.abs(P1 - P1.T).max() < 1e-18
    P2 = po.cov_clone(P, 0, 6)
    idx = list(range(n)) + list(range(6))
    assert np.array_equal(P2, P[np.ix_(idx, idx)])
    P3 = po.cov_clone(P, 9, 3)
    idx = list(range(n)) + [9, 10, 11]
    assert np.array_equal(P3, P[np.ix_(idx, idx)])


def test_cpi_records_are_the_preintegrated_trajectory(pkg):
    po = oracle_lib.load_prop(pkg)
    t, wm, am = synth.imu_stream(traj, 50.0, 50.2, rate=400.0)
    nz = pkg.imu_noise()
    imu = imu_at(pkg, 50.0)
    R0, p0 = traj(50.0)
    v0 = vel(50.0)
    acc = po.reset_cpi(imu, 50.0)
    acc_lib = pkg.reset_cpi(imu, 50.0)
    assert bytes(acc) == bytes(acc_lib) and acc.R_k2tau[0] == 1 and list(acc.v_clone) == list(imu.v)
    # one call per IMU message, as SystemManager::feed_measurement_imu drives it
    recs = []
    for i in range(len(t) - 1):
        recs += po.propagate(imu, nz, t[i:i + 2], wm[i:i + 2], am[i:i + 2], acc=acc)[2]
    for r in recs[::7] + recs[-1:]:
        Rk, pk = traj(r.t)
        dt = r.t - 50.0
        assert abs(r.dt - dt) < 1e-12 and r.clone_t == 50.0
        assert np.abs(np.array(r.R_I0toIk).reshape(3, 3) - Rk @ R0.T).max() < 1e-6      # midpoint rule on w, 400 Hz
        assert np.abs(np.array(r.alpha) - R0 @ (pk - p0 - v0 * dt + 0.5 * G * dt * dt)).max() < 1e-6
        Q = np.array(r.Q).reshape(6, 6)
        assert np.abs(Q - Q.T).max() < 1e-20 and np.linalg.eigvalsh(Q).min() > -1e-20 and Q[0, 0] > 0 and Q[3, 3] > 0
    # growth of the measurement covariance: gyro white noise integrates to sigma_w^2 * T on the rotation block
    Qn = np.array(recs[-1].Q).reshape(6, 6)
    assert abs(Qn[0, 0] / (nz.sigma_w ** 2 * 0.2) - 1) < 1e-2
    # the same stream in one call gives the same accumulator (and the reference's quirk on v only shows with several steps)
    imu2, acc2 = imu_at(pkg, 50.0), po.reset_cpi(imu_at(pkg, 50.0), 50.0)
    recs2 = po.propagate(imu2, nz, t, wm, am, acc=acc2)[2]
    assert np.abs(np.array(acc2.alpha_tau) - np.array(acc.alpha_tau)).max() < 1e-14
    assert np.abs(np.array(recs2[-1].R_I0toIk) - np.array(recs[-1].R_I0toIk)).max() < 1e-14
    assert np.abs(np.array(recs2[0].v) - np.array(recs[0].v)).max() < 1e-14


def test_cpi_integrate_oracle_and_clone_choice(pkg):
    """create_new_cpi_integrate: forward it lands on the records Propagator::propagate makes; backward (clone after the
    requested time) it still gives the pose of the trajectory through get_interpolated_pose_imu's formula."""
    po = oracle_lib.load_prop(pkg)
    t, wm, am = synth.imu_stream(traj, 50.0, 50.4, rate=400.0)
    nz = pkg.imu_noise()
    imu = imu_at(pkg, 50.1)
    Rc, pc = traj(50.1)
    vc = vel(50.1)
    # forward, ending on an IMU sample: the same numbers as the propagate chain started at the clone
    acc = po.reset_cpi(imu, 50.1)
    i0 = int(np.argmin(np.abs(t - 50.1)))
    _, st_, sw_, sa_ = po.select_imu_readings(t, wm, am, t[i0], t[i0 + 20])   # (drops the sample before time1, as the reference)
    assert len(st_) == 20
    recs = po.propagate(imu_at(pkg, 50.1), nz, st_, sw_, sa_, acc=acc)[2]
    ok, r = po.cpi_integrate(nz, t[i0 + 20], t[i0], Rc, vc, (0, 0, 0), (0, 0, 0), t, wm, am)
    assert ok and abs(r.dt - (t[i0 + 20] - t[i0])) < 1e-15
    assert np.abs(np.array(r.R_I0toIk) - np.array(recs[-1].R_I0toIk)).max() < 1e-14
    assert np.abs(np.array(r.alpha) - np.array(recs[-1].alpha)).max() < 1e-14
    assert np.abs(np.array(r.Q) - np.array(recs[-1].Q)).max() <= 1e-12 * np.abs(np.array(r.Q)).max()
    # between samples, forward and backward: p = p0 + v0 dt - g dt^2 / 2 + R0^T alpha is the trajectory
    for tq in (50.1 + 0.0437, 50.1 - 0.0612):
        ok, r = po.cpi_integrate(nz, tq, 50.1, Rc, vc, (0, 0, 0), (0, 0, 0), t, wm, am)
        assert ok and abs(r.dt - (tq - 50.1)) < 1e-15 and r.clone_t == 50.1
        Rq, pq = traj(tq)
        assert np.abs(np.array(r.R_I0toIk).reshape(3, 3) @ Rc - Rq).max() < 1e-6
        p = pc + vc * r.dt - 0.5 * G * r.dt ** 2 + Rc.T @ np.array(r.alpha)
        assert np.abs(p - pq).max() < 1e-6
    # outside the IMU buffer
    assert not po.cpi_integrate(nz, 49.9, 50.1, Rc, vc, (0, 0, 0), (0, 0, 0), t, wm, am)[0]
    # the clone choice (intent of closest_clone_time_not_imu)
    sc = synth.vio_scene(n_clones=6, F=4, M=5)
    st, _ = synth.scene_views(pkg, sc)
    ct = sc["t"]
    assert pkg.closest_clone_time(st, ct[2] + 0.01) == ct[2] and pkg.closest_clone_time(st, ct[2] + 0.03) == ct[3]
    assert pkg.closest_clone_time(st, ct[5] + 1.0) == ct[5] and pkg.closest_clone_time(st, ct[5] + 1.0, exclude_newest=True) == ct[4]


def test_next_clone_time(pkg):
    """SystemManager::get_next_clone_time (REF: SystemManager.cpp:172-267) against a direct Python restatement."""
    def ref(nc, st, mt, new, new2, is_imu, f, times, sdt, io, inew, wheel):
        if nc == 0:
            return st
        ct = (new2 if is_imu else new) + 1.0 / f
        ct = mt if ct < st else ct
        mind, tmp, have = np.inf, ct, False
        for x in reversed(times):
            s = x + sdt
            have = have or new < s
            if s < ct - 0.1 / f:
                break
            if abs(s - ct) < mind and s >= st:
                mind, tmp = abs(s - ct), s
        if inew < tmp or io > tmp:
            return None
        if np.isinf(mind) and not have and not wheel:
            return None
        return tmp

    rng = np.random.default_rng(3)
    cam = 10.0 + 0.05 * np.arange(40) + rng.uniform(-0.004, 0.004, 40)
    hits = 0
    for trial in range(300):
        new = 10.0 + rng.uniform(0, 1.5)
        args = dict(nc=int(rng.integers(0, 3)), st=new + rng.uniform(0, 0.12), mt=new + rng.uniform(0.0, 0.2), new=new,
                    new2=new - 0.1, is_imu=bool(rng.integers(0, 2)), f=int(rng.choice([10, 20])), times=cam[cam < new + rng.uniform(0, 0.3)],
                    sdt=rng.choice([0.0, 0.003]), io=9.0, inew=new + rng.uniform(0.0, 0.4), wheel=bool(rng.integers(0, 2)))
        got = pkg.next_clone_time(args["nc"], args["st"], args["mt"], args["new"], args["new2"], args["is_imu"], args["f"], args["times"],
                                  args["sdt"], args["io"], args["inew"], args["wheel"])
        exp = ref(**args)
        assert (got is None) == (exp is None) and (got is None or got == exp), (trial, got, exp)
        hits += got is not None
    assert 60 < hits < 300
    # snapping: a camera frame 2 ms after the desired time wins over the desired time itself
    got = pkg.next_clone_time(3, 10.0, 10.0, 10.0, 9.9, False, 10, [9.9, 10.0, 10.102], 0.0, 9.0, 10.2)
    assert got == 10.102
