"""tests/decision_trace.py on hand-made records (no device): the module the replay tests and profiles/r04/replay_vs_cpu_*.json rest on
must itself say the right thing about identical runs, a threshold tie, a real disagreement and a pool that differs."""
import numpy as np

import decision_trace as dt

THR = dict(max_cond=10000.0, min_dist=0.1, max_dist=150.0, max_baseline=40.0, reproj_px=3.0, res_norm=3.0)


def _vals(chi2, thr=7.81, cond=50.0, depth=8.0, reproj=0.4, n_obs=10):
    v = np.full(11, np.nan)
    v[dt.N_OBS], v[dt.TRI_OK], v[dt.REPROJ], v[dt.GATE_OK] = n_obs, 1, reproj, 1
    v[dt.COND], v[dt.DEPTH], v[dt.REF_DEPTH], v[dt.BASELINE] = cond, depth, depth, 5.0
    v[dt.CHI2], v[dt.CHI2_THR], v[dt.RES_NORM] = chi2, thr, 1.0
    return v


def _points(frame, ids, chi2s, status=0, thr=7.81):
    ids = np.array(ids, dtype=np.uint64)
    vals = np.stack([_vals(c, thr) for c in chi2s])
    acc = np.array([c < thr for c in chi2s], dtype=np.uint8)
    return ("points", frame, 0.05 * frame, len(ids), ids, acc, status, (ids, vals), np.full(6, 1e-3))


def test_identical_runs():
    a = [_points(f, [10 * f + 1, 10 * f + 2, 10 * f + 3], [1.0, 2.0, 9.0]) for f in range(1, 30)]
    b = [_points(f, [10 * f + 1, 10 * f + 2, 10 * f + 3], [1.0, 2.0, 9.0]) for f in range(1, 30)]
    s = dt.summary(a, b, thr=THR)
    assert s["updates"] == 29 and s["updates_with_identical_decisions"] == 29
    assert s["first_divergence"] is None and s["differing_updates"] == [] and s["tie_check"] == []


def test_a_value_on_its_threshold_is_a_tie():
    a = [_points(f, [10 * f + 1, 10 * f + 2], [1.0, 2.0]) for f in range(1, 20)]
    # (the runs start 1e-9 apart — the arithmetic alone — and have drifted to 1e-7 by the time a value lands on its threshold)
    b = [_points(f, [10 * f + 1, 10 * f + 2], [1.0 * (1 + (1e-9 if f <= 5 else 1e-7)), 2.0 * (1 - (1e-9 if f <= 5 else 1e-7))]) for f in range(1, 20)]
    a.append(_points(20, [201, 202], [1.0, 7.81 * (1 - 2e-7)]))        # accepted on one side ...
    b.append(_points(20, [201, 202], [1.0, 7.81 * (1 + 2e-7)]))        # ... rejected on the other: 4e-7 apart, on the threshold
    s = dt.summary(a, b, thr=THR)
    fd = s["first_divergence"]
    assert s["updates_with_identical_decisions"] == 19 and fd["update"] == 19 and fd["split"] == "chi2" and fd["frame"] == 20
    assert [int(i) for i in fd["ids"]["hip_only"]] == [202] and fd["n_accepted"] == {"hip": 2, "cpu": 1}
    t = fd["tie"][0]
    assert t["id"] == 202 and t["test"] == "chi2" and t["hip"]["passed"] and not t["cpu"]["passed"] and t["margin"] < 1e-6
    assert s["tie_check"] == [], s["tie_check"]
    d = s["differing_updates"]
    assert len(d) == 1 and d[0]["update"] == 19 and d[0]["accepted_a_only"] == [202] and d[0]["ids_a_only"] == []


def test_a_real_disagreement_is_not_a_tie():
    a = [_points(f, [10 * f + 1, 10 * f + 2], [1.0, 2.0]) for f in range(1, 20)]
    b = [_points(f, [10 * f + 1, 10 * f + 2], [1.0, 2.0]) for f in range(1, 20)]
    a.append(_points(20, [201, 202], [1.0, 3.0]))
    b.append(_points(20, [201, 202], [1.0, 30.0]))                     # a factor of ten apart: not rounding
    s = dt.summary(a, b, thr=THR)
    assert s["first_divergence"]["split"] == "chi2" and s["tie_check"], s
    assert "202" in s["tie_check"][0]


def test_a_pool_that_differs_is_reported_with_the_feature():
    a = [_points(f, [10 * f + 1, 10 * f + 2], [1.0, 2.0]) for f in range(1, 10)]
    b = [_points(f, [10 * f + 1, 10 * f + 2], [1.0, 2.0]) for f in range(1, 10)]
    a.append(_points(10, [101, 102, 77], [1.0, 2.0, 1.5]))             # feature 77 enters the pool a frame earlier in one run
    b.append(_points(10, [101, 102], [1.0, 2.0]))
    a.append(_points(11, [111, 112], [1.0, 2.0]))
    b.append(_points(11, [111, 112, 77], [1.0, 2.0, 1.5]))
    s = dt.summary(a, b, thr=THR)
    assert s["first_divergence"]["split"] == "pool" and s["first_divergence"]["detail"] == {"hip": 3, "cpu": 2}
    assert s["tie_check"] and "pool" in s["tie_check"][0]
    d = s["differing_updates"]
    assert [x["update"] for x in d] == [9, 10] and d[0]["ids_a_only"] == [77] and d[1]["ids_b_only"] == [77]
    assert s["updates_with_identical_decisions"] == 9


def test_min_unit_pivot():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((12, 12))
    P = A @ A.T + 12 * np.eye(12)
    d = np.sqrt(np.diag(P))
    L = np.linalg.cholesky(P / np.outer(d, d))
    # pivots of the unit-diagonal LDL^T are the squares of the Cholesky factor's diagonal
    assert abs(dt.min_unit_pivot(P) - (np.diag(L) ** 2).min()) < 1e-12
    # scaling the states changes nothing; a state that nearly repeats another shows as its conditional variance
    S = np.diag(10.0 ** rng.uniform(-3, 3, 12))
    assert abs(dt.min_unit_pivot(S @ P @ S) - dt.min_unit_pivot(P)) < 1e-10
    J = np.vstack([np.eye(12), np.eye(12)[3]])
    Q = J @ P @ J.T
    Q[12, 12] += 1e-9 * P[3, 3]
    assert 0.3e-9 < dt.min_unit_pivot(Q) < 1.1e-9
    Q[12, 12] -= 3e-9 * P[3, 3]
    assert dt.min_unit_pivot(Q) < 0
