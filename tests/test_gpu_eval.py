"""SURVEY §8(f) rank 1 on the GPU: plv_traj_ate (Umeyama passes + per-pose errors on the device) against the numpy
restatement, the committed toy fixture and size-independent properties on a long trajectory."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import eval_oracle as eo  # noqa: E402

pytestmark = pytest.mark.gpu
METHODS = ("posyaw", "posyawsingle", "se3", "se3single", "sim3", "none")


def _check(r, ref, tol=1e-11):
    assert np.abs(r["R"] - ref["R"]).max() < tol and np.abs(r["t"] - ref["t"]).max() < tol * 10 and abs(r["s"] - ref["s"]) < tol
    assert np.abs(r["pos_err"] - ref["pos_err"]).max() < 1e-10
    # acos near 1 amplifies rounding: orientation errors agree to ~1e-6 deg where the error itself is ~0
    assert np.abs(r["ori_err"] - ref["ori_err"]).max() < 1e-6
    for k in ref["pos"]:
        assert abs(r["pos"][k] - ref["pos"][k]) < 1e-10, k
        assert abs(r["ori"][k] - ref["ori"][k]) < 1e-6, k


def test_ate_matches_fixture_and_oracle(ctx):
    with open(os.path.join(ROOT, "tests", "golden", "ate_toy.json")) as f:
        d = json.load(f)
    gt, est = np.array(d["gt"]), np.array(d["est"])
    for m in METHODS:
        r = ctx.traj_ate(est, gt, m)
        ref = d["results"][m]
        assert np.abs(r["R"] - np.array(ref["R"])).max() < 1e-11 and np.abs(r["t"] - np.array(ref["t"])).max() < 1e-10
        assert np.abs(r["pos_err"] - np.array(ref["pos_err"])).max() < 1e-10
        assert np.abs(r["ori_err"] - np.array(ref["ori_err"])).max() < 1e-8
        for k, v in ref["pos"].items():
            assert abs(r["pos"][k] - v) < 1e-10, (m, k)
        o = eo.calculate_ate(est, gt, m)
        assert np.abs(r["aligned"] - o["aligned"]).max() < 1e-10
    assert np.abs(ctx.traj_ate(est, gt, "posyaw", n_aligned=1)["R"] - ctx.traj_ate(est, gt, "posyawsingle")["R"]).max() == 0


@pytest.mark.parametrize("n", [3, 257, 20000])
def test_ate_sizes_and_properties(ctx, n):
    from make_ate_toy import transform
    rng = np.random.default_rng(n)
    s = np.linspace(0, 60, n)
    gt = np.zeros((n, 7))
    gt[:, 0], gt[:, 1], gt[:, 2] = 30 * np.cos(0.2 * s), 20 * np.sin(0.3 * s), 0.5 * s
    for i in range(n):
        w = np.array([0.05 * np.sin(s[i]), 0.05 * np.cos(s[i]), 0.4 * s[i]])
        th = np.linalg.norm(w) + 1e-300
        K = eo.skew(w / th)
        gt[i, 3:] = eo.rot_2_quat((np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).T)
    Rz, t = eo.rot_z(-2.1), np.array([100.0, -50.0, 7.0])
    est = transform(gt, Rz.T, -Rz.T @ t)
    # exact rigid transform: every method that can represent it recovers it, errors vanish
    for m in ("posyaw", "se3", "sim3"):
        r = ctx.traj_ate(est, gt, m)
        assert np.abs(r["R"] - Rz).max() < 1e-10 and np.abs(r["t"] - t).max() < 1e-8 and abs(r["s"] - 1) < 1e-10
        assert r["pos_err"].max() < 1e-8 and r["ori_err"].max() < 1e-4
    est[:, :3] += rng.normal(0, 0.1, (n, 3))
    if n <= 257:
        for m in METHODS:
            _check(ctx.traj_ate(est, gt, m), eo.calculate_ate(est, gt, m))
    else:  # vectorised numpy for the long case: alignment from the oracle, errors recomputed in bulk
        r = ctx.traj_ate(est, gt, "se3")
        R, tt, _ = eo.align_trajectory(est, gt, "se3")
        assert np.abs(r["R"] - R).max() < 1e-11 and np.abs(r["t"] - tt).max() < 1e-9
        pe = np.linalg.norm(gt[:, :3] - (est[:, :3] @ R.T + tt), axis=1)
        assert np.abs(r["pos_err"] - pe).max() < 1e-9
        assert abs(r["pos"]["rmse"] - np.sqrt((pe ** 2).mean())) < 1e-10 and abs(r["pos"]["median"] - np.median(pe)) < 1e-10
        # invariance: moving both trajectories by one rigid motion changes no error
        R2, t2 = eo.rot_z(0.9) @ np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.0]]), np.array([5.0, 6.0, 7.0])
        r2 = ctx.traj_ate(transform(est, R2, t2), transform(gt, R2, t2), "se3")
        assert np.abs(r2["pos_err"] - r["pos_err"]).max() < 1e-8


def test_planar_trajectory_rank_deficient(ctx):
    """Ground vehicle: z == 0 makes the correlation matrix rank 2; U S V^T must still be the proper rotation."""
    n = 500
    s = np.linspace(0, 40, n)
    gt = np.zeros((n, 7))
    gt[:, 0], gt[:, 1] = s * np.cos(0.1 * s), s * np.sin(0.1 * s)
    gt[:, 6] = 1.0
    from make_ate_toy import transform
    Rz, t = eo.rot_z(1.2), np.array([3.0, 4.0, 0.0])
    est = transform(gt, Rz.T, -Rz.T @ t)
    for m in ("se3", "sim3", "posyaw"):
        r = ctx.traj_ate(est, gt, m)
        assert abs(np.linalg.det(r["R"]) - 1) < 1e-12 and np.abs(r["R"] - Rz).max() < 1e-10 and r["pos_err"].max() < 1e-9
