"""a19, use_imu_res branch: the oracle's get_interpolated_pose_imu / create_new_cpi_linear against the analytic
trajectory and an independent numpy / scipy restatement (REF: PL-VIWO/src/state/State.cpp:273-355,1138-1155)."""
import numpy as np
from scipy.spatial.transform import Rotation, Slerp

import oracle_lib
import synth


def _setup(pkg, **kw):
    sc = synth.vio_scene(n_clones=8, F=4, **kw)
    cp = synth.cpi_scene(sc)
    st, _ = synth.scene_views(pkg, sc)
    tab = pkg.CpiTable(cp["t"], cp["clone_t"], cp["R"], cp["alpha"], cp["v"], gravity=cp["gravity"])
    return sc, cp, st, tab


def numpy_pose(sc, cp, tq):
    """Independent restatement from the ORIGINAL table (no insertion)."""
    t = cp["t"]
    e = np.flatnonzero(t == tq)
    if len(e):
        i = int(e[0])
        Rk, al, ct = cp["R"][i], cp["alpha"][i], cp["clone_t"][i]
    else:
        i1 = int(np.searchsorted(t, tq, side="right"))
        i0 = i1 - 1
        lam = (tq - t[i0]) / (t[i1] - t[i0])
        Rk = Slerp([0, 1], Rotation.from_matrix([cp["R"][i0], cp["R"][i1]]))([lam]).as_matrix()[0]
        al = (1 - lam) * cp["alpha"][i0] + lam * cp["alpha"][i1]
        ct = cp["clone_t"][i0]
    ci = int(np.flatnonzero(sc["t"] == ct)[0])
    vi = int(np.flatnonzero(t == ct)[0])
    dt = tq - ct
    R0, p0 = sc["R"][ci], sc["p"][ci]
    return Rk @ R0, p0 + cp["v"][vi] * dt - 0.5 * cp["gravity"] * dt * dt + R0.T @ al


def test_cpi_pose_at_records_is_the_trajectory(pkg):
    jo = oracle_lib.load_jac(pkg)
    sc, cp, st, tab = _setup(pkg)
    tq = cp["t"][::3]
    R, p, ok = jo.cpi_poses(st, tab, tq)
    assert ok.all()
    for q, tt in enumerate(tq):
        Rt, pt = sc["pose_fn"](tt) if tt not in sc["t"] else (sc["R"][list(sc["t"]).index(tt)], sc["p"][list(sc["t"]).index(tt)])
        assert np.abs(R[q].reshape(3, 3) - Rt).max() < 1e-12
        assert np.abs(p[q] - pt).max() < 1e-9  # v0 comes from a central difference


def test_cpi_linear_interpolation(pkg):
    jo = oracle_lib.load_jac(pkg)
    sc, cp, st, tab = _setup(pkg)
    rng = np.random.default_rng(1)
    inner = cp["t"][:-1][np.diff(cp["clone_t"]) == 0]            # left ends of intervals inside one clone's run
    tq = inner[rng.integers(0, len(inner), 40)] + rng.uniform(0.0005, 0.0045, 40)
    R, p, ok = jo.cpi_poses(st, tab, tq)
    assert ok.all()
    for q, tt in enumerate(tq):
        Rn, pn = numpy_pose(sc, cp, tt)
        assert np.abs(R[q].reshape(3, 3) - Rn).max() < 1e-12 and np.abs(p[q] - pn).max() < 1e-12
        Rt, pt = sc["pose_fn"](tt)
        # linear alpha misses the curvature g dt^2 / 2 inside a 5 ms step: up to 9.81 * 0.005^2 / 8 = 3e-5 m
        assert np.abs(R[q].reshape(3, 3) - Rt).max() < 1e-5 and np.abs(p[q] - pt).max() < 1e-4
    # the reference stores what it interpolates: asking again, and asking between an interpolated and an original
    # record, stays on the same geodesic / line
    t_a = inner[5] + 0.002
    t_b = inner[5] + 0.001
    R2, p2, ok2 = jo.cpi_poses(st, tab, np.array([t_a, t_b, t_a]))
    assert ok2.all() and np.array_equal(R2[0], R2[2]) and np.array_equal(p2[0], p2[2])
    Rn, pn = numpy_pose(sc, cp, t_b)
    assert np.abs(R2[1].reshape(3, 3) - Rn).max() < 1e-12 and np.abs(p2[1] - pn).max() < 1e-12


def test_cpi_failures(pkg):
    jo = oracle_lib.load_jac(pkg)
    sc, cp, st, tab = _setup(pkg)
    t = cp["t"]
    # out of the table; between the last record of one clone's run and the next clone's record (different clones)
    last_of_run = t[:-1][np.diff(cp["clone_t"]) != 0]
    tq = np.array([t[0] - 0.01, t[-1] + 0.01, last_of_run[2] + 1e-4])
    R, p, ok = jo.cpi_poses(st, tab, tq)
    assert not ok.any() and not R.any() and not p.any()
    # a window that has lost its two oldest clones: records integrated from them are unusable, exact or interpolated
    st2 = pkg.StateView(sc["t"][2:], sc["R"][2:], sc["p"][2:], sc["ids"][2:], sc["R_ItoC"], sc["p_IinC"], sc["K8"])
    R, p, ok = jo.cpi_poses(st2, tab, np.array([t[1], t[1] + 0.002, sc["t"][3], sc["t"][3] + 0.0125]))
    assert list(ok) == [0, 0, 1, 1]
    # empty table
    empty = pkg.CpiTable(np.zeros(0), np.zeros(0), np.zeros((0, 9)), np.zeros((0, 3)), np.zeros((0, 3)))
    assert not jo.cpi_poses(st, empty, np.array([sc["t"][1]]))[2].any()
