"""SURVEY §8(f) rank 1 on the CPU: the numpy restatement of ov_eval against the committed toy fixture and hand-computed
transforms, and the host-only entry points of the library (writer, loader, association) against the restatement."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import eval_oracle as eo  # noqa: E402


@pytest.fixture(scope="module")
def toy():
    with open(os.path.join(ROOT, "tests", "golden", "ate_toy.json")) as f:
        d = json.load(f)
    d["gt"], d["est"] = np.array(d["gt"]), np.array(d["est"])
    return d


def test_oracle_reproduces_fixture_and_known_transform(toy):
    for m, ref in toy["results"].items():
        r = eo.calculate_ate(toy["est"], toy["gt"], m)
        assert np.abs(r["R"] - np.array(ref["R"])).max() < 1e-12 and np.abs(r["pos_err"] - np.array(ref["pos_err"])).max() < 1e-12
        for k, v in ref["pos"].items():
            assert abs(r["pos"][k] - v) < 1e-12, (m, k)
    # 5 cm / 1 deg of noise around a yaw of 0.7 rad and t = (1.5, -2, 0.3): the estimate of the transform is close to it
    r = eo.calculate_ate(toy["est"], toy["gt"], "posyaw")
    assert np.abs(r["R"] - eo.rot_z(toy["yaw"])).max() < 0.01 and np.abs(r["t"] - np.array(toy["t"])).max() < 0.05
    assert 0.05 < r["pos"]["rmse"] < 0.15
    # Umeyama's optimum: no other yaw / translation does better on the position RMSE
    base = r["pos"]["rmse"]
    for dyaw, dt in ((0.01, 0), (-0.01, 0), (0, 0.02)):
        R2 = eo.rot_z(np.arctan2(r["R"][1, 0], r["R"][0, 0]) + dyaw)
        p2 = (R2 @ toy["est"][:, :3].T).T + r["t"] + dt
        assert np.sqrt(((toy["gt"][:, :3] - p2) ** 2).sum(1).mean()) >= base - 1e-12


def test_sim3_recovers_scale(toy):
    gt = toy["gt"]
    est = gt.copy()
    est[:, :3] = (gt[:, :3] - np.array([1.0, 2.0, 3.0])) / 1.25   # gt = 1.25 * est + (1, 2, 3)
    r = eo.calculate_ate(est, gt, "sim3")
    assert abs(r["s"] - 1.25) < 1e-12 and np.abs(r["R"] - np.eye(3)).max() < 1e-12 and np.abs(r["t"] - [1, 2, 3]).max() < 1e-12
    assert r["pos_err"].max() < 1e-12


def test_writer_loader_roundtrip(pkg, toy, tmp_path):
    P = np.diag([1e-4, 2e-4, 3e-4, 1e-2, 2e-2, 3e-2])
    P[0, 1] = P[1, 0] = 1.23456789e-5
    P[3, 5] = P[5, 3] = -4.5e-3
    path = tmp_path / "traj.txt"
    with open(path, "w") as f:
        f.write(pkg.traj_header())
        assert pkg.traj_header() == eo.HEADER
        for i, pose in enumerate(toy["est"]):
            line = pkg.traj_format(1403636579.763555 + 0.05 * i, pose[:3], pose[3:], P if i % 2 == 0 else None)
            assert line == eo.format_pose(1403636579.763555 + 0.05 * i, pose[:3], pose[3:], P if i % 2 == 0 else None)
            f.write(line)
        f.write("#trailing comment\n \n1.0 2.0 3.0\n")  # comment, blank, short line: all skipped
    t, poses, co, cp = pkg.traj_load(path)
    t_o, poses_o, co_o, cp_o = eo.load_data(str(path))
    assert len(t) == len(toy["est"]) == len(t_o) and len(co) == (len(t) + 1) // 2 == len(co_o)
    assert np.array_equal(t, t_o) and np.array_equal(poses, poses_o) and np.array_equal(co, co_o) and np.array_equal(cp, cp_o)
    assert np.abs(poses - toy["est"]).max() <= 5.1e-7 and abs(t[1] - t[0] - 0.05) < 2e-6   # precision 6
    assert abs(co[0][0, 1] - 1.23456789e-5) < 1e-10 and co[0][1, 0] == co[0][0, 1] and cp[0][2, 0] == cp[0][0, 2]
    assert abs(pkg.traj_length(poses) - eo.total_length(poses_o)) < 1e-12
    with pytest.raises(pkg.PlvError):
        pkg.traj_load(tmp_path / "missing.txt")
    empty = tmp_path / "empty.txt"
    empty.write_text("# nothing\n")
    with pytest.raises(pkg.PlvError):
        pkg.traj_load(empty)


def test_association(pkg):
    rng = np.random.default_rng(4)
    gt_t = 100.0 + 0.01 * np.arange(2000) + rng.uniform(-0.002, 0.002, 2000)
    est_t = np.sort(np.concatenate([100.3 + 0.05 * np.arange(300) + rng.uniform(-0.01, 0.01, 300), [90.0, 130.0]]))
    for off, md in ((0.0, 0.02), (0.013, 0.02), (0.0, 0.003)):
        ei, gi = pkg.traj_associate(est_t, gt_t, off, md)
        eo_i, go_i = eo.perform_association(off, md, est_t, gt_t)
        assert np.array_equal(ei, eo_i) and np.array_equal(gi, go_i)
        assert (np.abs(gt_t[gi] - est_t[ei] - off) <= md).all() and (np.diff(gi) > 0).all()
    ei, gi = pkg.traj_associate(est_t, gt_t)
    assert len(ei) == 300 and 0 not in ei
    # duplicated estimate stamps: the ground-truth pointer never goes back, so the second one pairs with the next sample
    ei, gi = pkg.traj_associate(np.array([100.5, 100.5, 100.5]), gt_t, 0.0, 0.02)
    eo_i, go_i = eo.perform_association(0.0, 0.02, np.array([100.5, 100.5, 100.5]), gt_t)
    assert np.array_equal(ei, eo_i) and np.array_equal(gi, go_i) and len(set(gi)) == len(gi)
    assert len(pkg.traj_associate(np.zeros(0), gt_t)[0]) == 0
