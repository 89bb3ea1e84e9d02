"""SURVEY §8(f) rank 3 on the GPU: the wheel system against the oracle, and UpdaterWheel::update (gate + EKF update with the
full 6x6 preintegrated covariance) against the textbook formulas in numpy."""
import numpy as np
import pytest

import oracle_lib
import synth
from test_oracle_wheel import make, wheel_stream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,ext,dt,intr", [(0, False, False, False), (0, True, True, True), (1, True, False, False), (2, False, True, False),
                                              (3, True, True, True), (4, False, False, False), (5, True, True, False)])
def test_wheel_linear_system_parity(pkg, kind, ext, dt, intr):
    po = oracle_lib.load_prop(pkg)
    ctx = pkg.Context(pkg.default_config(752, 480))
    t, m1, m2 = wheel_stream(20.0, 21.0, kind=kind)
    rng = np.random.default_rng(kind)
    m1, m2 = m1 + rng.normal(0, 0.05, m1.shape), m2 + rng.normal(0, 0.05, m2.shape)
    ok, st_, s1, s2 = pkg.select_wheel_data(t, m1, m2, 20.3007, 20.8004)
    assert ok
    d0, d1 = rng.normal(0, 0.01, 6), rng.normal(0, 0.01, 6)
    opt, st = make(pkg, kind, ext, dt, intr, t0=20.3007, t1=20.8004, d0=d0, d1=d1)
    st.R0_fej[1] += 1e-3   # first estimates that differ from the values
    st.p1_fej[0] -= 2e-3
    H, res, Cov, cols, R3, p3 = ctx.wheel_linear_system(opt, st, st_, s1, s2)
    Ho, reso, Covo, colso, R3o, p3o = po.wheel_linear_system(opt, st, st_, s1, s2)
    assert H.shape == Ho.shape and np.array_equal(cols, colso)
    assert np.abs(H - Ho).max() < 1e-12 * max(1.0, np.abs(Ho).max()) and np.abs(res - reso).max() < 1e-12
    assert np.abs(Cov - Covo).max() < 1e-12 * np.abs(Covo).max() and np.abs(Cov - Cov.T).max() == 0
    assert H.shape[0] == (3 if kind >= 3 else 6) and Cov.shape == (H.shape[0], H.shape[0])
    assert np.abs(R3 - R3o).max() < 1e-13 and np.abs(p3 - p3o).max() < 1e-12
    ctx.close()


def test_wheel_update_is_the_full_R_ekf_update(pkg):
    po = oracle_lib.load_prop(pkg)
    ctx = pkg.Context(pkg.default_config(752, 480))
    n = 15 + 6 * 4
    t, m1, m2 = wheel_stream(20.0, 21.0)
    ok, st_, s1, s2 = pkg.select_wheel_data(t, m1, m2, 20.3, 20.8)
    rng = np.random.default_rng(4)
    q95 = synth.q95_table()
    for kind, scale, expect in ((0, 0.02, 1), (0, 3.0, 0), (3, 0.02, 1), (5, 3.0, 0)):     # clone errors inside / far outside the gate
        d0, d1 = rng.normal(0, scale * 0.05, 6), rng.normal(0, scale * 0.05, 6)
        if kind != 0:
            t, m1, m2 = wheel_stream(20.0, 21.0, kind=kind)
            ok, st_, s1, s2 = pkg.select_wheel_data(t, m1, m2, 20.3, 20.8)
        opt, st = make(pkg, kind, ext=True, intr=(kind % 3 == 0), d0=d0, d1=d1)
        P = synth.spd_cov(n, seed=8) * 1e-3
        ctx.cov_upload(P)
        rc, acc, dx = ctx.wheel_update(opt, st, st_, s1, s2, n)
        H, res, Cov, cols, _, _ = po.wheel_linear_system(opt, st, st_, s1, s2)
        Hf = np.zeros((len(res), n))
        Hf[:, cols] = H
        S = Hf @ P @ Hf.T + Cov
        chi2 = res @ np.linalg.solve(S, res)
        if kind == 0:
            assert (chi2 < opt.chi2_mult * q95[len(res)]) == bool(expect)
        expect = chi2 < opt.chi2_mult * q95[len(res)]      # (the 2D noise model is wide: both of its cases pass the gate)
        assert bool(expect) == bool(acc) and rc == 0, (kind, chi2)
        Pd = ctx.cov_download(n)
        if expect:
            K = P @ Hf.T @ np.linalg.inv(S)
            assert np.abs(dx - K @ res).max() < 1e-9 * max(1.0, np.abs(K @ res).max())
            assert np.abs(Pd - (P - K @ Hf @ P)).max() < 1e-9 * np.abs(P).max()
        else:
            assert not dx.any() and np.array_equal(Pd, P)
    # a state the columns do not fit
    opt, st = make(pkg, 0)
    st.pose1_id = n
    with pytest.raises(pkg.PlvError):
        ctx.wheel_update(opt, st, st_, s1, s2, n)
    ctx.close()
