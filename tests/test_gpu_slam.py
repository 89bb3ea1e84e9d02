"""GPU parity of the SLAM-landmark entry points (a30) against the oracle, through the C-ABI."""
import numpy as np
import pytest

import oracle_lib
import synth
from test_oracle_slam import landmark_system

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k,rows,seed", [(40, 24, 12, 3), (113, 98, 30, 5), (60, 18, 4, 8), (143, 128, 24, 9)])
def test_slam_initialize_parity(ctx, oracle, n, k, rows, seed):
    q95 = synth.q95_table()
    P, cols, Hf, Hx, res = landmark_system(n, k, rows, seed)
    ok_o, P2_o, dxi_o, dx_o = oracle.slam_initialize(P, Hf, Hx, res, cols, q95, chi2_mult=5.0)
    ctx.cov_upload(P)
    ok, dxi, dx = ctx.slam_initialize(n, Hf, Hx, res, cols, chi2_mult=5.0)
    assert ok == ok_o == 1
    P2 = ctx.cov_download(n + 3)
    assert np.abs(P2 - P2_o).max() <= 1e-9 * np.abs(P2_o).max()
    assert np.abs(dxi - dxi_o).max() <= 1e-9 * max(1.0, np.abs(dxi_o).max())
    assert np.abs(dx - dx_o).max() <= 1e-8 * max(1e-3, np.abs(dx_o).max())
    # the landmark now is an ordinary state: update it with a fresh measurement, then marginalise it again
    rng = np.random.default_rng(seed)
    cols2 = np.concatenate([cols[:12], [n, n + 1, n + 2]]).astype(np.int32)
    H = rng.normal(size=(8, len(cols2)))
    r = rng.normal(0, 0.4, 8)
    rc_o, P3_o, acc_o, dxu_o = oracle.slam_update(P2_o, H, r, cols2, q95, chi2_mult=5.0)
    rc, acc, dxu = ctx.slam_update(n + 3, H, r, cols2, chi2_mult=5.0)
    assert rc == rc_o == 0 and acc == acc_o == 1
    assert np.abs(ctx.cov_download(n + 3) - P3_o).max() <= 1e-9 * np.abs(P3_o).max()
    assert np.abs(dxu - dxu_o).max() <= 1e-8 * max(1e-3, np.abs(dxu_o).max())
    ctx.cov_marginalize(n, 3)
    assert np.abs(ctx.cov_download(n) - oracle.cov_marginalize(P3_o, n, 3)).max() <= 1e-9 * np.abs(P3_o).max()


def test_slam_initialize_rejections_leave_state(ctx, oracle):
    q95 = synth.q95_table()
    n = 40
    P, cols, Hf, Hx, res = landmark_system(n, 24, 12, 4)
    ctx.cov_upload(P)
    bad = res.copy()
    bad[5:] += 80.0
    assert ctx.slam_initialize(n, Hf, Hx, bad, cols)[0] == 0          # Mahalanobis gate
    assert ctx.slam_initialize(n, Hf, Hx, np.zeros_like(res), cols)[0] == 0   # chi < 1e-7
    assert np.array_equal(ctx.cov_download(n), P)
    rc, acc, dx = ctx.slam_update(n, Hx, res + 60.0, cols)
    assert acc == 0 and not dx.any() and np.array_equal(ctx.cov_download(n), P)
