"""Two runs of the replay driver compared decision by decision (test infrastructure).

The driver records one entry per camera update (plviwo_amd.system.SystemManager(decisions=[...])): which features / lines of the pool
were triangulated and which of those passed the gate.  The front-ends of the HIP library and of the CPU oracle are bit-identical, so
the two filters see the same measurements and their decisions can only part where a floating-point test value sits on its
threshold.  first_divergence() finds the first update whose decisions differ and says which test split them:

    pool            the pools differ in size — measurements differ: not a tie this module can put a margin on (seen once in 13 480
                    updates of a 300 s drive with the intrinsics calibrated online: the tracker's RANSAC undistorts with the state's
                    intrinsics, so a track can end a frame apart in the two runs; summary()'s differing_updates names the feature)
    triangulation   an id was triangulated by one side only (condition number / depth range / baseline ratio of
                    FeatureInitializer::single_triangulation, single_gaussnewton; LineHelper's triangulation for lines)
    chi2            an id was triangulated by both and accepted by one only (UpdaterStatistics::Chi2Check, the residual-norm gate;
                    for a wheel update: its one chi2 test)
    status          the update as a whole was rejected by one side only

and tie_margin() puts a number on it: the test value's relative distance to its threshold, evaluated by the CPU oracle on the state of
each run (the states agree to rounding before the first divergence, so a tie shows as a tiny margin on both sides, a real
disagreement as an O(1) one)."""
import numpy as np


# columns of Context.last_point_decisions / FrameOracle.last_point_decisions
N_OBS, TRI_OK, REPROJ, GATE_OK, COND, DEPTH, REF_DEPTH, BASELINE, CHI2, CHI2_THR, RES_NORM = range(11)


def tests_of(v, thr):
    """The tests one pool entry went through, in the reference's order: [(name, value, threshold, passed)] up to the first it failed.
    thr: dict(max_cond, min_dist, max_dist, max_baseline, reproj_px, res_norm)."""
    out = []

    def add(name, value, limit, passed):
        out.append((name, float(value), float(limit), bool(passed)))
        return passed

    def depth(name, z):
        lim = thr["min_dist"] if abs(z - thr["min_dist"]) < abs(z - thr["max_dist"]) else thr["max_dist"]
        return add(name, z, lim, thr["min_dist"] <= z <= thr["max_dist"])
    if np.isnan(v[COND]):
        return out
    if not add("condition number of the linear triangulation", v[COND], thr["max_cond"], v[COND] <= thr["max_cond"]) or not depth("depth of the linear triangulation", v[DEPTH]):
        return out
    if not np.isnan(v[REF_DEPTH]):
        if not depth("depth after the refinement", v[REF_DEPTH]) or not add("baseline ratio", v[BASELINE], thr["max_baseline"], v[BASELINE] <= thr["max_baseline"]):
            return out
    if np.isnan(v[REPROJ]) or not add("mean reprojection error, px", v[REPROJ], thr["reproj_px"], v[REPROJ] < thr["reproj_px"]):
        return out
    if np.isnan(v[CHI2]):
        return out
    if not add("norm of the projected residual", v[RES_NORM], thr["res_norm"], v[RES_NORM] < thr["res_norm"]):
        return out
    add("chi2", v[CHI2], v[CHI2_THR], v[CHI2] < v[CHI2_THR])
    return out


def line_tests_of(v):
    """a line entry's gate (REF UpdaterCamera.cpp:406-419: chi2 against chi2_mult x the 95 % quantile; no residual-norm gate): values =
    chi2, threshold, residual norm as plv_last_line_decisions returns them"""
    if np.isnan(v[0]) or np.isnan(v[1]):
        return []
    return [("chi2 of the line", float(v[0]), float(v[1]), bool(v[0] < v[1]))]


def tie_record(fid, rec_a, rec_b, thr, names=("hip", "cpu")):
    """The test that split the two runs on feature `fid` and its margin — |value - threshold| / |threshold| — on both sides."""
    sides = {}
    lines = rec_a[0] == "lines"
    for name, rec in zip(names, (rec_a, rec_b)):
        if rec[7] is None:
            return dict(id=int(fid), note=f"{name}: no values recorded")
        ids, vals = rec[7]
        i = np.nonzero(ids == np.uint64(fid))[0]
        if len(i) == 0:
            return dict(id=int(fid), note=f"{name}: the feature is not in the pool")
        sides[name] = line_tests_of(vals[i[0]]) if lines else tests_of(vals[i[0]], thr)
    ta, tb = sides[names[0]], sides[names[1]]
    for (na, va, la, pa), (nb, vb, lb, pb) in zip(ta, tb):
        if pa != pb:
            ma, mb = abs(va - la) / abs(la), abs(vb - lb) / abs(lb)
            return dict(id=int(fid), test=na, **{names[0]: dict(value=va, threshold=la, passed=pa, margin=ma),
                                                 names[1]: dict(value=vb, threshold=lb, passed=pb, margin=mb)},
                        margin=max(ma, mb), values_differ_by=abs(va - vb) / max(abs(va), abs(vb), 1e-300))
    return dict(id=int(fid), note="every recorded test agrees", tests={names[0]: ta, names[1]: tb})


def first_divergence(a, b, names=("hip", "cpu"), thr=None):
    """a, b: decision lists of two runs.  Returns None when every decision agrees, else a dict describing the first difference (with
    thr — the thresholds of tests_of — and recorded values: the test that split the runs and its margins, `tie`)."""
    d = _first_divergence(a, b, names)
    if d is not None and thr is not None and d.get("kind") in ("points", "lines") and "ids" in d:
        k = d["update"]
        d["tie"] = [tie_record(fid, a[k], b[k], thr, names) for fid in sorted(set(d["ids"][names[0] + "_only"]) | set(d["ids"][names[1] + "_only"]))]
    return d


def _first_divergence(a, b, names):
    for k, (ra, rb) in enumerate(zip(a, b)):
        kind, frame, t, pool_a, ids_a, acc_a, st_a = ra[:7]
        kind_b, frame_b, t_b, pool_b, ids_b, acc_b, st_b = rb[:7]
        rec = dict(update=k, kind=kind, frame=int(frame), state_time=float(t))
        if kind != kind_b or frame != frame_b:
            return dict(rec, split="sequence", detail=f"{names[0]}: {kind} of frame {frame}, {names[1]}: {kind_b} of frame {frame_b}")
        if pool_a != pool_b:
            return dict(rec, split="pool", detail={names[0]: pool_a, names[1]: pool_b})
        sa, sb = set(int(i) for i in ids_a), set(int(i) for i in ids_b)
        if sa != sb:
            return dict(rec, split="triangulation", ids={names[0] + "_only": sorted(sa - sb), names[1] + "_only": sorted(sb - sa)},
                        n_triangulated={names[0]: len(sa), names[1]: len(sb)})
        if st_a != st_b:
            return dict(rec, split="status", detail={names[0]: st_a, names[1]: st_b})
        da = {int(i): int(x) for i, x in zip(ids_a, acc_a)}
        db = {int(i): int(x) for i, x in zip(ids_b, acc_b)}
        diff = sorted(i for i in da if da[i] != db[i])
        if diff:
            return dict(rec, split="chi2", ids={names[0] + "_only": [i for i in diff if da[i]], names[1] + "_only": [i for i in diff if db[i]]},
                        n_accepted={names[0]: sum(da.values()), names[1]: sum(db.values())})
    if len(a) != len(b):
        return dict(update=min(len(a), len(b)), split="sequence", detail=f"{len(a)} updates against {len(b)}")
    return None


def value_drift(a, b, thr):
    """How far apart the two runs' test values are, update by update: for every point update both runs recorded, the largest relative
    difference of a recorded value (condition number, depths, baseline ratio, reprojection error, chi2, residual norm) over the
    features that pass every test in both runs (a failing entry's values are often garbage: the depth of a point at infinity) —
    [(update, largest difference, which value, feature id)].  The first updates show the arithmetic alone (same state in, two
    implementations, values of condition 1e3 .. 1e4); later ones the two filters' states drifting apart."""
    names = ("n_obs", "triangulated", "reproj_px", "gate_passed", "tri_cond", "tri_depth", "refined_depth", "baseline_ratio", "chi2", "chi2_threshold", "res_norm")
    cols = [2, 4, 5, 6, 7, 8, 10]
    out = []
    for k, (ra, rb) in enumerate(zip(a, b)):
        if ra[0] != "points" or rb[0] != "points" or ra[7] is None or rb[7] is None:
            continue
        (ia, va), (ib, vb) = ra[7], rb[7]
        common, xa, xb = np.intersect1d(ia, ib, return_indices=True)
        keep = [q for q in range(len(common)) if all(len(t) == 7 and all(x[3] for x in t) for t in (tests_of(va[xa[q]], thr), tests_of(vb[xb[q]], thr)))]
        if not keep:
            continue
        A, B = va[xa[keep]][:, cols], vb[xb[keep]][:, cols]
        rel = np.abs(A - B) / np.maximum(np.maximum(np.abs(A), np.abs(B)), 1e-300)
        i, j = np.unravel_index(np.argmax(rel), rel.shape)
        out.append((k, float(rel[i, j]), names[cols[j]], int(common[keep[i]])))
    return out


def check_tie(summary_, first_updates=3, first_tol=1e-8, window=10, factor=20.0):
    """What the replay tests assert about a divergence.  (1) It is a decision on values (triangulation / chi2), never on the
    measurements (pool), the sequence of updates or an update's status.  (2) The two implementations agree on identical input: over
    the first updates — the states have not had time to part — every passing entry's values agree to first_tol.  (3) The entry that
    split the runs sits ON its threshold as far as the runs can tell: the threshold lies between the two values, and the two values
    are no further apart than `factor` x what passing entries of the `window` point updates before already differ by.  Returns the
    list of violations (empty: a tie)."""
    bad = []
    fd, drift = summary_["first_divergence"], summary_["value_drift"]
    early = [d for d in drift["all"] if d[0] < 2 * first_updates][:first_updates]
    if early and max(d[1] for d in early) > first_tol:
        bad.append(f"the first updates' values differ by {max(d[1] for d in early):.3g} (> {first_tol:g}): {early}")
    if fd is None:
        return bad
    if fd["split"] not in ("triangulation", "chi2"):
        return bad + [f"the runs part on '{fd['split']}', not on a test value: {fd}"]
    if fd["kind"] not in ("points", "lines"):
        return bad
    before = [d[1] for d in drift["all"] if d[0] < fd["update"]][-window:]
    allowed = max(1e-9, factor * max(before)) if before else 1e-9
    for t in fd.get("tie", []):
        if "test" not in t:
            bad.append(f"no deciding test found for feature {t['id']}: {t}")
        elif t["values_differ_by"] > allowed:
            bad.append(f"feature {t['id']}: '{t['test']}' differs by {t['values_differ_by']:.3g} between the runs, passing entries of the {window} "
                       f"updates before by at most {max(before):.3g}")
    return bad


def thresholds(op):
    """The thresholds of tests_of from loaded options (plviwo_amd.options): the feature initialiser's, the 3 px test of
    UpdaterCamera.cpp:656-683 and the residual-norm gate the update runs with."""
    fi = op.est.cam.featinit
    return dict(max_cond=fi.max_cond_number, min_dist=fi.min_dist, max_dist=fi.max_dist, max_baseline=fi.max_baseline, reproj_px=3.0, res_norm=3.0)


def summary(a, b, thr=None):
    """Counts over the whole of both runs: updates, updates that agree in every decision, the first divergence."""
    n = min(len(a), len(b))
    same = 0
    differing = []
    for k, (ra, rb) in enumerate(zip(a, b)):
        if (ra[0] == rb[0] and ra[3] == rb[3] and ra[6] == rb[6] and np.array_equal(ra[4], rb[4]) and np.array_equal(ra[5], rb[5])):
            same += 1
        elif len(differing) < 12:
            ia, ib = set(int(v) for v in ra[4]), set(int(v) for v in rb[4])
            acc_a = set(int(v) for v, f in zip(ra[4], ra[5]) if f)
            acc_b = set(int(v) for v, f in zip(rb[4], rb[5]) if f)
            differing.append(dict(update=k, kind=[ra[0], rb[0]], frame=[int(ra[1]), int(rb[1])], pool=[int(ra[3]), int(rb[3])], status=[int(ra[6]), int(rb[6])],
                                  ids_a_only=sorted(ia - ib)[:8], ids_b_only=sorted(ib - ia)[:8],
                                  accepted_a_only=sorted(acc_a - acc_b)[:8], accepted_b_only=sorted(acc_b - acc_a)[:8]))
    drift = value_drift(a, b, thr) if thr is not None else []
    fd = first_divergence(a, b, thr=thr)
    upto = fd["update"] if fd is not None and "update" in fd else n
    before = [d for d in drift if d[0] < upto]
    out = dict(updates=n, updates_with_identical_decisions=same, first_divergence=fd, differing_updates=differing,
                value_drift=dict(what="largest relative difference between the two runs' recorded test values per point update, over the entries that pass every test in both (update, difference, value, feature)",
                                 all=drift, first_updates=drift[:5], before_the_first_divergence=before[-5:],
                                 largest_before_the_first_divergence=max(before, key=lambda d: d[1]) if before else None,
                                 at_every_20th_update=drift[::20]),
                tie_check=None)
    out["tie_check"] = check_tie(out) if thr is not None else None
    # the first update whose pools agree and whose verdicts do not (a run that parted on a pool goes on: what splits it NEXT is the
    # informative event), with the deciding test of every entry the two sides judged differently and its margin on either side
    out["first_decision_on_values"] = None
    if thr is not None:
        for k, (ra, rb) in enumerate(zip(a, b)):
            if ra[0] != rb[0] or ra[3] != rb[3] or ra[6] != rb[6] or ra[0] not in ("points", "lines"):
                continue
            acc_a = {int(v): int(f) for v, f in zip(ra[4], ra[5])}
            acc_b = {int(v): int(f) for v, f in zip(rb[4], rb[5])}
            ids = sorted((set(acc_a) ^ set(acc_b)) | {i for i in set(acc_a) & set(acc_b) if acc_a[i] != acc_b[i]})
            if ids:
                out["first_decision_on_values"] = dict(update=k, kind=ra[0], frame=int(ra[1]), ids=ids[:8], tie=[tie_record(i, ra, rb, thr) for i in ids[:8]])
                break
    del out["value_drift"]["all"]
    return out


class ProbedTrace(list):
    """a decision list that also asks the driver (plviwo_amd.system.SystemManager) for the state and covariance in front of every
    camera update (states_pre: (frame, state vector, covariance)) and, with probe_cov = (first frame, last frame), for the covariance
    after every step of those frames"""
    probe_state = True
    probe_cov = None

    def __init__(self):
        super().__init__()
        self.states, self.states_pre, self.states_prop, self.cov_probes = [], [], [], []


def min_unit_pivot(P):
    """smallest pivot of the Cholesky factorisation (no pivoting) of the covariance scaled to unit diagonal: the conditional variance
    of the most dependent state given the ones before it, as a fraction of its variance (<= 0: not positive definite to rounding).
    Clone positions of this filter sit at 1e-9 .. 1e-8: the quantity an update's rounding is measured against."""
    d = np.sqrt(np.abs(np.diag(P)))
    A = P / np.outer(d, d)
    lo = 1.0
    for j in range(len(A)):
        piv = A[j, j]
        lo = min(lo, piv)
        if piv <= 1e-300:
            A[j + 1:, j] = 0.0
            continue
        A[j + 1:, j] /= piv
        A[j + 1:, j + 1:] -= np.outer(A[j + 1:, j], A[j + 1:, j]) * piv
    return lo
