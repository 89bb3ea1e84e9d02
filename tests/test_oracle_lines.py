"""CPU checks of the line-feature oracle (a27-a29): LineHelper restatement, oracle/jacobian_oracle.cpp."""
import numpy as np
import pytest

import oracle_lib
import synth


@pytest.fixture(scope="module")
def jo(pkg):
    return oracle_lib.load_jac(pkg)


def views(pkg, sc, ls, line_FinG=None, **kw):
    st, _ = synth.scene_views(pkg, sc, **kw)
    lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"],
                        line_FinG=ls["lines"] if line_FinG is None else line_FinG)
    return st, lt


def test_line_residual_vanishes_on_exact_projection(pkg, jo):
    sc = synth.vio_scene(F=4, calib_int=False)
    ls = synth.line_scene(sc, L=12, noise_px=0.0)
    st, lt = views(pkg, sc, ls)
    cols = jo.line_columns(st, lt)
    assert len(cols) == 6 * 15  # every clone, no calibration columns (the line model has none)
    rows, Hf, Hx, res = jo.build_line_jacobians(st, lt, cols, 32)
    assert (rows == 2 * np.diff(ls["obs_ptr"])).all()
    # end points lie on the projected line: point-line distance ~ float rounding of the pixels (whitened by 1/sigma)
    assert np.abs(res).max() < 5e-3
    assert np.isfinite(Hf).all() and np.isfinite(Hx).all()


def test_line_jacobian_column_blocks(pkg, jo):
    sc = synth.vio_scene(F=4, calib_int=False)
    ls = synth.line_scene(sc, L=6, M=5, noise_px=0.3)
    st, lt = views(pkg, sc, ls)
    cols = jo.line_columns(st, lt)
    rows, Hf, Hx, res = jo.build_line_jacobians(st, lt, cols, 32)
    # an observation at clone time t_i only touches the four clones of its interpolation window
    ids = list(sc["ids"])
    for l in range(lt.c.n_lines):
        r = rows[l]
        used = np.abs(Hx[l][:, :r]).max(axis=1) > 0
        touched = {cols[j] for j in np.nonzero(used)[0]}
        first = min(ids.index(c - (c - ids[0]) % 6) for c in touched)
        assert first >= len(ids) - 5 - 3  # window of the oldest of its 5 views starts at most 3 clones earlier


def test_line_triangulation_direction_and_moment(pkg, jo):
    # the reference rejects plane pairs closer than acos(0.99) = 8 deg: needs a wide baseline
    sc = synth.vio_scene(F=4, calib_int=False, dt_clone=0.5)
    ls = synth.line_scene(sc, L=20, noise_px=0.0, depth=(4.0, 12.0))
    st, lt = views(pkg, sc, ls)
    out, ok = jo.triangulate_lines(st, lt)
    assert ok.sum() >= 15  # near-degenerate plane pairs (|cos| >= 0.99 for every view) are rejected
    for l in np.nonzero(ok)[0]:
        v_true, n_true = ls["lines"][l, 3:], ls["lines"][l, :3]
        v, n = out[l, 3:], out[l, :3]
        c = abs(v @ v_true) / np.linalg.norm(v)
        assert c > 1 - 1e-6
        # the reference normalises the direction (sum / sum of norms) but only averages the moment
        # (LineHelper.cpp:462-465), so [n; v] is scale-consistent in the anchor camera frame only:
        # undo the frame change and compare the moment's direction there
        ci = 15 - (ls["obs_ptr"][l + 1] - ls["obs_ptr"][l])
        R0 = sc["R_ItoC"] @ sc["R"][ci]
        p0 = sc["p"][ci] - R0.T @ sc["p_IinC"]
        n_c0 = R0 @ (n - np.cross(p0, v))
        n_true_c0 = R0 @ (n_true - np.cross(p0, v_true))
        s = np.sign(v @ v_true)
        assert (n_c0 @ n_true_c0) * s / (np.linalg.norm(n_c0) * np.linalg.norm(n_true_c0)) > 1 - 1e-6


def test_line_triangulation_from_point_and_class(pkg, jo):
    sc = synth.vio_scene(F=4, calib_int=False)
    ls = synth.line_scene(sc, L=5, noise_px=0.0)
    st, _ = synth.scene_views(pkg, sc)
    pts = np.arange(15.0).reshape(5, 3) + 1
    lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], D=[1, 2, 3, 0, 2],
                        anchor_pt=pts, has_pt=[1, 1, 1, 1, 0])
    out, ok = jo.triangulate_lines(st, lt)
    assert ok[:3].all()
    for l, D in enumerate([1, 2, 3]):
        first_clone = 15 - (ls["obs_ptr"][l + 1] - ls["obs_ptr"][l])
        R = sc["R"][first_clone]
        d = R.T @ np.eye(3)[D - 1]
        assert np.allclose(out[l, 3:], d, atol=1e-12)
        assert np.allclose(out[l, :3], np.cross(pts[l], d), atol=1e-12)


def _gate(pkg, jo, sc, ls, st, lines):
    """the reference's lines_update gate (UpdaterCamera.cpp:404-419) on the lines given as Pluecker coordinates: accepted flags"""
    orc = oracle_lib.load()
    lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=lines)
    cols = jo.line_columns(st, lt)
    rows, Hf, Hx, res = jo.build_line_jacobians(st, lt, cols, 32)
    P = synth.spd_cov(sc["n_state"], seed=4) * 1e-4
    rc, P1, dx, acc, nrows = orc.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, synth.q95_table(), res_norm_gate=0.0)
    return np.asarray(acc, dtype=bool)


def test_why_the_reference_accepts_few_lines(pkg, jo):
    """VERDICT r5 item 5 — why a line update of this filter rarely reaches the EKF (bench.py: ~50 lines triangulated and < 1 accepted per
    frame, identically on the HIP library and on the CPU oracle).  Exact 3-D lines seen without pixel noise from a wide-baseline window:

    (a) handed to the gate with their TRUE Pluecker coordinates, (nearly) every one passes: the gate and the Jacobians are sound;
    (b) triangulated by the reference's own routine for unclassified lines (line_single_triangulation: plane pairs, LineHelper.cpp:372-
        470) most fail the same gate: the routine averages the pairs' directions normalised but their moments as they come out of the
        plane intersection (:462-465), so [n; v] carries a scale error of the order |n_0||n_1| sin(angle) — the line's distance from the
        origin is wrong by that factor and the residual of every view with it;
    (c) the route that can pass is a CLASSIFIED line (D > 0: parallel to a body axis by its vanishing point) through a triangulated
        point of its own (line_triangulation_from_points_and_direction, :231-290): n = p x v with v the body axis of its class — accepted
        when that axis really is the line's direction, rejected when it is not (the next test).
    On the rendered drives 70 % of the kept lines are unclassified (Vanishing_Points projects a body axis without dividing by its depth,
    LineHelper.cpp:1039-1046: only lines aiming at that pseudo vanishing point classify), and lines_update skips what has fewer than
    three usable views (res.size() < 5, UpdaterCamera.cpp:406) without returning it: what is left is what bench.py counts."""
    sc = synth.vio_scene(F=4, calib_int=False, dt_clone=0.5)
    ls = synth.line_scene(sc, L=40, noise_px=0.0, depth=(4.0, 12.0))
    st, lt = views(pkg, sc, ls)
    true_ok = _gate(pkg, jo, sc, ls, st, ls["lines"])
    assert true_ok.mean() > 0.9, true_ok.mean()                                          # (a)
    out, ok = jo.triangulate_lines(st, lt)
    assert ok.sum() >= 30
    tri = np.where(ok[:, None], out, ls["lines"])
    tri_ok = _gate(pkg, jo, sc, ls, st, tri)
    frac = tri_ok[ok].mean()
    print("lines accepted by the gate: true coordinates %.0f %%, the reference's plane-pair triangulation %.0f %%" % (100 * true_ok.mean(), 100 * frac))
    assert frac < 0.35, frac                                                                 # (b)
    # the scale error itself: |n| of the triangulated line against the true one, in the anchor camera's frame (where the routine works)
    ratio = []
    for l in np.nonzero(ok)[0]:
        ci = 15 - (ls["obs_ptr"][l + 1] - ls["obs_ptr"][l])
        R0 = sc["R_ItoC"] @ sc["R"][ci]
        p0 = sc["p"][ci] - R0.T @ sc["p_IinC"]
        n_c0 = R0 @ (out[l, :3] - np.cross(p0, out[l, 3:]))
        n_true = R0 @ (ls["lines"][l, :3] - np.cross(p0, ls["lines"][l, 3:]))
        ratio.append(np.linalg.norm(n_c0) / np.linalg.norm(n_true))
    ratio = np.array(ratio)
    assert np.median(np.abs(np.log(ratio))) > np.log(1.5), np.median(ratio)              # off by far more than any noise would put it


def test_a_classified_line_passes_when_its_class_is_right(pkg, jo):
    """(c) above: n = p x v with v the body axis of the line's class (LineHelper.cpp:249-283).  The six-column null-space projection
    removes the line's own error to first order, so an anchor 5 px off the line (the tolerance of AssignPointToLines) still passes the
    gate; what does not pass is a line whose class is wrong — v off by 24 degrees, the slant of the boulevard scene's courses, which the
    pseudo vanishing point classifies as 'along the drive' — the error is then far outside the linearisation."""
    sc = synth.vio_scene(F=4, calib_int=False, dt_clone=0.5)
    ls = synth.line_scene(sc, L=40, noise_px=0.0, depth=(4.0, 12.0))
    st, _ = views(pkg, sc, ls)
    v = ls["lines"][:, 3:]
    a = np.cross(v, ls["lines"][:, :3])          # the line's point closest to the origin (|v| = 1)
    on = np.concatenate([np.cross(a, v), v], axis=1)
    assert np.allclose(on, ls["lines"], atol=1e-9)
    acc_on = _gate(pkg, jo, sc, ls, st, on)
    off, slant = [], []
    th = np.deg2rad(24.0)
    for l in range(len(v)):
        view = a[l] - sc["p"][-1]
        side = np.cross(v[l], view)
        side /= np.linalg.norm(side)
        p = a[l] + side * 5.0 / sc["K8"][0] * np.linalg.norm(view)      # across the line, 5 px at the depth of the line
        off.append(np.concatenate([np.cross(p, v[l]), v[l]]))
        v2 = np.cos(th) * v[l] + np.sin(th) * side                        # the direction turned by 24 degrees about the viewing ray
        slant.append(np.concatenate([np.cross(a[l], v2), v2]))
    acc_off, acc_slant = _gate(pkg, jo, sc, ls, st, np.array(off)), _gate(pkg, jo, sc, ls, st, np.array(slant))
    print("classified lines accepted: anchor on the line %.0f %%, anchor 5 px off %.0f %%, direction 24 deg off %.0f %%"
          % (100 * acc_on.mean(), 100 * acc_off.mean(), 100 * acc_slant.mean()))
    assert acc_on.mean() > 0.9 and acc_off.mean() > 0.8 and acc_slant.mean() < 0.2, (acc_on.mean(), acc_off.mean(), acc_slant.mean())
