"""est.use_imu_cov on the CPU: plv_cpi_noise (host arithmetic of the library: the CPI record behind a query time, REF State.cpp:273-355)
and the oracle's noise branch (REF CamHelper.cpp:217-224)."""
import numpy as np

import oracle_lib
import synth


def _table(pkg, rng, clone_t, n=9):
    t = np.concatenate([[c + 0.01 * i for i in range(n)] for c in clone_t])
    ct = np.repeat(clone_t, n)
    A = rng.normal(0, 1e-2, (len(t), 6, 6))
    Q = (A @ np.transpose(A, (0, 2, 1))).reshape(len(t), 36)
    R = np.tile(np.eye(3).ravel(), (len(t), 1))
    return pkg.CpiTable(t, ct, R, np.zeros((len(t), 3)), np.zeros((len(t), 3)), Q=Q), t, ct, Q


def test_cpi_noise_lookup(pkg):
    sc = synth.vio_scene(n_clones=6, F=4, M=4)
    st, _ = synth.scene_views(pkg, sc)
    rng = np.random.default_rng(1)
    clone_t = np.array(sc["t"][1:4])
    tab, t, ct, Q = _table(pkg, rng, clone_t)
    tq = np.array([t[3], 0.5 * (t[3] + t[4]), t[0] - 1.0, t[-1] + 1.0, 0.25 * t[12] + 0.75 * t[13]])
    Qo, ci, ok = pkg.cpi_noise(st, tab, tq)
    assert list(ok) == [1, 1, 0, 0, 1]
    assert np.array_equal(Qo[0], Q[3]) and ci[0] == 1                       # a stored record: its own covariance, clone index in the view
    assert np.allclose(Qo[1], 0.5 * (Q[3] + Q[4]), rtol=1e-9, atol=1e-15)      # create_new_cpi_linear: (1 - lambda) Q0 + lambda Q1
    assert np.allclose(Qo[4], 0.25 * Q[12] + 0.75 * Q[13], rtol=1e-7, atol=1e-14) and ci[4] == 2
    assert not Qo[2].any() and ci[2] == -1
    # between the last record of one clone and the first of the next: different clones -> no record (REF :334-338)
    gap = 0.5 * (t[8] + t[9])
    if t[8] < gap < t[9]:
        assert pkg.cpi_noise(st, tab, [gap])[2][0] == 0


def test_oracle_noise_branch_is_linear_in_q_times_mlt(pkg):
    jo = oracle_lib.load_jac(pkg)
    sc = synth.vio_scene(n_clones=8, F=6, M=6, obs_offset=0.011)
    rng = np.random.default_rng(3)
    nobs = len(sc["obs_time"])
    res_R = np.array([sc["pose_fn"](t)[0] for t in sc["obs_time"]])
    res_p = np.array([sc["pose_fn"](t)[1] for t in sc["obs_time"]])
    A = rng.normal(0, 5e-3, (nobs, 6, 6))
    Q = (A @ np.transpose(A, (0, 2, 1))).reshape(nobs, 36)
    ci = rng.integers(0, len(sc["t"]), nobs).astype(np.int32)

    def run(use, mlt, q):
        st, _ = synth.scene_views(pkg, sc, use_imu_cov=use, intr_err_mlt=mlt)
        tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], sc["pts"], res_R=res_R, res_p=res_p, res_Q=q, res_clone=ci)
        return jo.build_jacobians(st, tr, jo.columns(st, tr), 16)

    base, zero, a, b = run(0, 1.0, Q), run(1, 1.0, 0 * Q), run(1, 1.0, Q), run(1, 0.25, 4 * Q)
    for x, y in zip(base[1:], zero[1:]):
        assert np.array_equal(x, y)                       # Q = 0: the plain sigma_pix^2 noise
    # (the reference whitens with the Cholesky factor of the SYMMETRIC matrix built from chol(R)'s entries, CamHelper.cpp:227-229: with a
    # strongly correlated R that matrix is indefinite and the rows are NaN, in the reference as here)
    for x, y in zip(a[1:], b[1:]):
        assert np.allclose(x, y, rtol=1e-12, atol=1e-15, equal_nan=True)  # only Q * mlt enters
    assert np.nansum(np.abs(a[2])) < np.nansum(np.abs(base[2]))         # and it makes the whitened rows smaller
