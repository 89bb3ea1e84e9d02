"""Writes tests/golden/ate_toy.json: a toy trajectory, a hand-chosen rigid transform of it plus fixed perturbations, and
the alignment / ATE numbers of oracle/eval_oracle.py on that pair (SURVEY §8(c) fixture viii).  The transform itself is
the known answer: with zero perturbation every method must recover it exactly (checked here before writing).
Run from the repo root:  python tests/golden/make_ate_toy.py"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import eval_oracle as eo  # noqa: E402


def toy(n=40):
    poses = np.zeros((n, 7))
    for i in range(n):
        s = 0.25 * i
        poses[i, :3] = [4.0 * math.cos(0.3 * s), 3.0 * math.sin(0.3 * s), 0.2 * s + 0.1 * math.sin(s)]
        w = np.array([0.02 * math.sin(s), 0.03 * math.cos(0.5 * s), 0.3 * s])
        th = np.linalg.norm(w)
        K = eo.skew(w / th) if th > 0 else np.zeros((3, 3))
        R_ItoG = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K
        poses[i, 3:] = eo.rot_2_quat(R_ItoG.T)  # JPL: quat of R_GtoI
    return poses


def transform(poses, R, t, s=1.0):
    out = poses.copy()
    q = eo.rot_2_quat(R)
    for i in range(len(poses)):
        out[i, :3] = s * R @ poses[i, :3] + t
        out[i, 3:] = eo.quat_multiply(poses[i, 3:], eo.quat_inv(q))
    return out


def main():
    gt = toy()
    yaw = 0.7
    Rz, t = eo.rot_z(yaw), np.array([1.5, -2.0, 0.3])
    est_exact = transform(gt, Rz.T, -Rz.T @ t)  # est = inverse transform of gt, so that est -> gt is (Rz, t)
    for m in ("posyaw", "se3", "sim3"):
        r = eo.calculate_ate(est_exact, gt, m)
        assert np.abs(r["R"] - Rz).max() < 1e-12 and np.abs(r["t"] - t).max() < 1e-12 and abs(r["s"] - 1) < 1e-12, m
        assert r["pos_err"].max() < 1e-12 and r["ori_err"].max() < 1e-5, m
    rng = np.random.default_rng(8)
    est = est_exact.copy()
    est[:, :3] += rng.normal(0, 0.05, (len(est), 3))
    for i in range(len(est)):
        dq = np.concatenate([rng.normal(0, 0.01, 3), [1.0]])
        est[i, 3:] = eo.quat_multiply(dq / np.linalg.norm(dq), est[i, 3:])
    out = dict(gt=gt.tolist(), est=est.tolist(), yaw=yaw, t=t.tolist(), results={})
    for m in ("posyaw", "posyawsingle", "se3", "se3single", "sim3", "none"):
        r = eo.calculate_ate(est, gt, m)
        out["results"][m] = dict(R=r["R"].tolist(), t=r["t"].tolist(), s=r["s"], ori=r["ori"], pos=r["pos"],
                                 ori_err=r["ori_err"].tolist(), pos_err=r["pos_err"].tolist())
    with open(os.path.join(ROOT, "tests", "golden", "ate_toy.json"), "w") as f:
        json.dump(out, f)
    print("wrote ate_toy.json; posyaw rmse pos %.6f m ori %.6f deg" % (out["results"]["posyaw"]["pos"]["rmse"], out["results"]["posyaw"]["ori"]["rmse"]))


if __name__ == "__main__":
    main()
