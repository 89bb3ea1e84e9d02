"""SURVEY §8(f) rank 2 on the GPU: plv_propagate / plv_cov_clone against the oracle through the C-ABI, and one
clone-window cycle (propagate -> augment_clone -> marginalize_old_clone) on the resident covariance."""
import numpy as np
import pytest

import oracle_lib
import synth
from test_oracle_propagate import imu_at, traj

pytestmark = pytest.mark.gpu


def _rec_eq(a, b, tol=1e-12):
    for f in ("t", "dt", "clone_t"):
        assert abs(getattr(a, f) - getattr(b, f)) <= 1e-15 * max(1.0, abs(getattr(b, f)))
    for f, scale in (("R_I0toIk", 1.0), ("alpha", 1.0), ("v", 10.0), ("w", 1.0)):
        assert np.abs(np.array(getattr(a, f)) - np.array(getattr(b, f))).max() < tol * scale, f
    qa, qb = np.array(a.Q), np.array(b.Q)
    assert np.abs(qa - qb).max() <= 1e-10 * np.abs(qb).max() + 1e-30


@pytest.mark.parametrize("n,imu_id,steps", [(113, 0, 11), (63, 12, 2), (15, 0, 41)])
def test_propagate_parity(pkg, n, imu_id, steps):
    po = oracle_lib.load_prop(pkg)
    ctx = pkg.Context(pkg.default_config(752, 480))
    bg, ba = (0.01, -0.02, 0.005), (0.05, 0.02, -0.03)
    t, wm, am = synth.imu_stream(traj, 50.0, 50.0 + (steps - 1) / 200.0, rate=200.0, bg=bg, ba=ba)
    rng = np.random.default_rng(steps)
    wm, am = wm + rng.normal(0, 1e-3, wm.shape), am + rng.normal(0, 1e-2, am.shape)
    nz = pkg.imu_noise()
    P = synth.spd_cov(n, seed=2) * 1e-3
    imu_o, imu_d = imu_at(pkg, 50.0, bg, ba), imu_at(pkg, 50.0, bg, ba)
    # first estimates that differ from the value, as after an update
    for s in (imu_o, imu_d):
        s.p_fej[0] += 0.01
        s.v_fej[1] -= 0.02
    acc_o, acc_d = po.reset_cpi(imu_o, 50.0), pkg.reset_cpi(imu_d, 50.0)
    Phi_o, Qd_o, rec_o, P_o = po.propagate(imu_o, nz, t, wm, am, P=P, acc=acc_o, imu_id=imu_id)
    ctx.cov_upload(P)
    Phi_d, Qd_d, rec_d = ctx.propagate(imu_d, nz, t, wm, am, n, acc=acc_d, imu_id=imu_id)
    assert np.abs(imu_d.vec() - imu_o.vec()).max() < 1e-12 and list(imu_d.q) == list(imu_d.q_fej)
    assert np.abs(Phi_d - Phi_o).max() < 1e-12 * np.abs(Phi_o).max()
    assert np.abs(Qd_d - Qd_o).max() < 1e-12 * np.abs(Qd_o).max() and np.abs(Qd_d - Qd_d.T).max() == 0
    assert len(rec_d) == len(rec_o) == steps - 1
    for a, b in zip(rec_d, rec_o):
        _rec_eq(a, b)
    assert abs(acc_d.DT - acc_o.DT) < 1e-15 and np.abs(np.array(acc_d.P_meas) - np.array(acc_o.P_meas)).max() <= 1e-10 * np.abs(np.array(acc_o.P_meas)).max()
    Pd = ctx.cov_download(n)
    # (the IMU block Phi P Phi^T + Q is not re-symmetrised by EKFPropagation: rounding-level asymmetry, as in the reference)
    assert np.abs(Pd - P_o).max() < 1e-12 * np.abs(P_o).max() and np.abs(Pd - Pd.T).max() < 1e-15 * np.abs(Pd).max()
    # message by message, as SystemManager::feed_measurement_imu drives it, with the accumulator carried across calls
    imu_s, acc_s = imu_at(pkg, 50.0, bg, ba), None
    for s in (imu_s,):
        s.p_fej[0] += 0.01
        s.v_fej[1] -= 0.02
    acc_s = pkg.reset_cpi(imu_s, 50.0)
    ctx.cov_upload(P)
    imu_so, acc_so, P_so = imu_at(pkg, 50.0, bg, ba), None, P
    imu_so.p_fej[0] += 0.01
    imu_so.v_fej[1] -= 0.02
    acc_so = po.reset_cpi(imu_so, 50.0)
    for i in range(min(steps - 1, 6)):
        _, _, r_d = ctx.propagate(imu_s, nz, t[i:i + 2], wm[i:i + 2], am[i:i + 2], n, acc=acc_s, imu_id=imu_id)
        _, _, r_o, P_so = po.propagate(imu_so, nz, t[i:i + 2], wm[i:i + 2], am[i:i + 2], P=P_so, acc=acc_so, imu_id=imu_id)
        _rec_eq(r_d[0], r_o[0])
    assert np.abs(ctx.cov_download(n) - P_so).max() < 1e-12 * np.abs(P_so).max()
    # without the CPI accumulator and without a covariance (mean + Phi only)
    imu_m, imu_mo = imu_at(pkg, 50.0), imu_at(pkg, 50.0)
    Phi_m, _, rec_m = ctx.propagate(imu_m, nz, t, wm, am, 0)
    Phi_mo = po.propagate(imu_mo, nz, t, wm, am)[0]
    assert rec_m == [] and np.abs(Phi_m - Phi_mo).max() < 1e-12 and np.abs(imu_m.vec() - imu_mo.vec()).max() < 1e-12
    ctx.close()


def test_clone_window_cycle(pkg, oracle):
    """propagate to the clone time -> augment_clone (StateHelper::clone of the IMU pose) -> marginalize the oldest clone."""
    po = oracle_lib.load_prop(pkg)
    ctx = pkg.Context(pkg.default_config(752, 480))
    n0 = 15 + 6 * 4
    P = synth.spd_cov(n0, seed=6) * 1e-3
    ctx.cov_upload(P)
    nz = pkg.imu_noise()
    t, wm, am = synth.imu_stream(traj, 50.0, 50.05, rate=200.0)
    imu_d, imu_o = imu_at(pkg, 50.0), imu_at(pkg, 50.0)
    ctx.propagate(imu_d, nz, t, wm, am, n0)
    P1 = po.propagate(imu_o, nz, t, wm, am, P=P)[3]
    ctx.cov_clone(n0, 0, 6)
    P2 = po.cov_clone(P1, 0, 6)
    got = ctx.cov_download(n0 + 6)
    assert np.abs(got - P2).max() < 1e-12 * np.abs(P2).max()
    assert np.array_equal(got[n0:, n0:], got[:6, :6]) and np.array_equal(got[n0:, :n0], got[:6, :n0])
    ctx.cov_marginalize(15, 6)   # the oldest clone sits right after the IMU block
    P3 = oracle.cov_marginalize(P2, 15, 6)
    assert np.abs(ctx.cov_download(n0) - P3).max() < 1e-12 * np.abs(P3).max()
    with pytest.raises(pkg.PlvError):
        ctx.cov_clone(n0 + 1, 0, 6)   # stale n
    ctx.close()


def test_cpi_integrate_parity_and_pose(pkg):
    """a19's last stage: plv_cpi_integrate against the oracle, then the record through plv_cpi_poses."""
    po, jo = oracle_lib.load_prop(pkg), oracle_lib.load_jac(pkg)
    ctx = pkg.Context(pkg.default_config(752, 480))
    t, wm, am = synth.imu_stream(traj, 50.0, 50.4, rate=400.0)
    rng = np.random.default_rng(5)
    wm, am = wm + rng.normal(0, 1e-3, wm.shape), am + rng.normal(0, 1e-2, am.shape)
    nz = pkg.imu_noise()
    Rc, pc = traj(50.2)
    vc = (traj(50.2 + 1e-5)[1] - traj(50.2 - 1e-5)[1]) / 2e-5
    bg, ba = (0.001, -0.002, 0.0005), (0.01, 0.02, -0.01)
    recs = []
    for tq in (50.2 + 0.0437, 50.2 - 0.0612, t[100], 50.2 + 0.0001):
        ok_o, r_o = po.cpi_integrate(nz, tq, 50.2, Rc, vc, bg, ba, t, wm, am)
        ok_d, r_d = ctx.cpi_integrate(nz, tq, 50.2, Rc, vc, bg, ba, t, wm, am)
        assert ok_o and ok_d
        _rec_eq(r_d, r_o)
        recs.append(r_d)
    assert not ctx.cpi_integrate(nz, 49.0, 50.2, Rc, vc, bg, ba, t, wm, am)[0]
    # the caller's State::cpis: the clone's own record + the new ones; plv_cpi_poses answers the requested times from them
    rows = sorted([(50.2, 50.2, np.eye(3).ravel(), np.zeros(3), vc)] + [(r.t, r.clone_t, np.array(r.R_I0toIk), np.array(r.alpha), np.array(r.v))
                                                                          for r in recs])
    tab = pkg.CpiTable([x[0] for x in rows], [x[1] for x in rows], [x[2] for x in rows], [x[3] for x in rows], [x[4] for x in rows])
    st = pkg.StateView([50.2], [Rc], [pc], [15], np.eye(3), np.zeros(3), synth.EUROC_K8)
    tq = np.array([r.t for r in recs])
    R, p, ok = ctx.cpi_poses(st, tab, tq)
    Ro, po_, oko = jo.cpi_poses(st, tab, tq)
    assert ok.all() and oko.all() and np.abs(R - Ro).max() < 1e-13 and np.abs(p - po_).max() < 1e-13
    for q, tt in enumerate(tq):   # noisy IMU: the pose is the trajectory up to the injected noise over <= 60 ms
        Rt, pt = traj(tt)
        assert np.abs(R[q].reshape(3, 3) - Rt).max() < 1e-3 and np.abs(p[q] - pt).max() < 1e-3
    ctx.close()
