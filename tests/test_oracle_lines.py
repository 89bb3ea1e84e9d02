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
