"""CPU checks of the SLAM-landmark oracle (a30): StateHelper::initialize / marginalize restatement."""
import numpy as np

import oracle_lib
import synth


def landmark_system(n, k, rows, seed, noise=0.3, prior=1e-3):
    rng = np.random.default_rng(seed)
    P = synth.spd_cov(n, seed=seed) * prior
    cols = synth.col_map(n, k, seed=seed + 1, skip=min(15, n - k))
    Hf = rng.normal(size=(rows, 3))
    Hx = rng.normal(size=(rows, k)) * 0.5
    x_true = rng.normal(size=3) * 0.4
    res = Hf @ x_true + rng.normal(0, noise, rows)
    return P, cols, Hf, Hx, res


def test_initialize_matches_dense_formulas():
    orc = oracle_lib.load()
    q95 = synth.q95_table()
    n, k, rows = 40, 24, 12
    P, cols, Hf, Hx, res = landmark_system(n, k, rows, 3)
    ok, P2, dxi, dx = orc.slam_initialize(P, Hf, Hx, res, cols, q95, chi2_mult=5.0)
    assert ok == 1
    # independent statement: QR of Hf splits the system; the new block is H_L^-1 (H_R P H_R^T + I) H_L^-T before the
    # update with the remaining rows, which can only shrink it
    Q, R = np.linalg.qr(Hf, mode="complete")
    HR = (Q.T @ Hx)[:3]
    Ps = P[np.ix_(cols, cols)]
    HLinv = np.linalg.inv(R[:3])
    PLL0 = HLinv @ (HR @ Ps @ HR.T + np.eye(3)) @ HLinv.T
    PLL = P2[n:, n:]
    assert np.allclose(PLL, PLL.T, atol=1e-12)
    ev = np.linalg.eigvalsh(PLL0 - PLL)
    assert ev.min() > -1e-9                     # information was added, never removed
    assert np.all(np.linalg.eigvalsh(P2) > 0)   # augmented covariance is SPD
    # the old block only shrinks as well
    assert np.linalg.eigvalsh(P - P2[:n, :n]).min() > -1e-12
    # Givens and QR agree up to row signs: |dx_init| matches the QR solve
    v = HLinv @ (Q.T @ res)[:3]
    assert np.allclose(np.abs(dxi), np.abs(v), atol=1e-9)


def test_initialize_rejections():
    orc = oracle_lib.load()
    q95 = synth.q95_table()
    P, cols, Hf, Hx, res = landmark_system(40, 24, 12, 4)
    # a gross outlier in the updating rows fails the Mahalanobis gate
    bad = res.copy()
    bad[5:] += 80.0
    assert orc.slam_initialize(P, Hf, Hx, bad, cols, q95)[0] == 0
    # a vanishing residual trips the "suspicious" test chi < 1e-7 (StateHelper.cpp:574)
    assert orc.slam_initialize(P, Hf, Hx, np.zeros_like(res), cols, q95)[0] == 0


def test_marginalize_is_block_removal():
    orc = oracle_lib.load()
    P = synth.spd_cov(20, seed=2)
    out = orc.cov_marginalize(P, 5, 3)
    keep = [i for i in range(20) if not 5 <= i < 8]
    assert np.array_equal(out, P[np.ix_(keep, keep)])


def test_slam_update_is_gated_ekf():
    orc = oracle_lib.load()
    q95 = synth.q95_table()
    n, k, rows = 40, 27, 10
    rng = np.random.default_rng(7)
    P = synth.spd_cov(n, seed=5) * 1e-3
    cols = synth.col_map(n, k, seed=6, skip=10)
    H = rng.normal(size=(rows, k))
    res = rng.normal(0, 0.5, rows)
    rc, P1, acc, dx = orc.slam_update(P, H, res, cols, q95)
    assert rc == 0 and acc == 1
    Hf = np.zeros((rows, n))
    Hf[:, cols] = H
    S = Hf @ P @ Hf.T + np.eye(rows)
    K = P @ Hf.T @ np.linalg.inv(S)
    assert np.allclose(P1, P - K @ Hf @ P, atol=1e-12) and np.allclose(dx, K @ res, atol=1e-12)
    rc, P2, acc, dx = orc.slam_update(P, H, res + 50.0, cols, q95)
    assert acc == 0 and np.array_equal(P2, P) and not dx.any()
