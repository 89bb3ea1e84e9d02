"""plv_camera_try_update / plv_camera_frame on hand-made databases: the point half of the one-call form against
plv_camera_update_points + a dx applied by the caller (same database, same covariance), and the state the library moved."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu

TRI = dict(max_cond=1e7, max_dist=100.0, max_baseline=1e3)


def _filled_context(pkg, sc, fo):
    ctx = pkg.Context(pkg.default_config(752, 480))
    for f in range(len(sc["obs_ptr"]) - 1):
        a, b = sc["obs_ptr"][f], sc["obs_ptr"][f + 1]
        uv = sc["obs_uv"][a:b].astype(np.float32)
        ctx.db_append_measurements(f + 1, sc["obs_time"][a:b].copy(), uv, fo.undistort(sc["K8"], uv))
    ctx.cov_upload(synth.spd_cov(sc["n_state"], seed=4) * 1e-4)
    return ctx


def test_try_update_points_half_and_the_state_it_moves(pkg):
    fo = oracle_lib.load_front()
    sc = synth.vio_scene(F=60, M=15, noise_px=0.4, seed=11)
    n, t = sc["n_state"], sc["t"]
    kw = dict(t_prev_frame=t[-2], state_time=t[-1], window_full=True, **TRI)
    # two-call form
    st_a, _ = synth.scene_views(pkg, sc)
    a = _filled_context(pkg, sc, fo)
    ref = a.camera_update_points(st_a, n, 40, 15, **kw)
    Pa = a.cov_download(n)
    # one-call form: the clone positions and the intrinsics of the view are variables the library moves
    st_b, _ = synth.scene_views(pkg, sc)
    p0, K0 = st_b.p.copy(), np.array(st_b.c.intrinsics)
    K = K0.copy()
    base = C.addressof(st_b.c)
    ent = [("vec", int(st_b.ids[i]) + 3, st_b.p[i], None, None) for i in range(len(st_b.ids))]
    ent.append(("vec", sc["intr_id"], K, None, base + pkg.PlvStateView.intrinsics.offset))
    plus = pkg.BoxPlus(ent)
    b = _filled_context(pkg, sc, fo)
    out, lines, n_db = b.camera_try_update(st_b, plus, n, 40, 15, lines=False, **kw)
    assert lines is None and n_db == 0
    assert out["n_accepted"] == ref["n_accepted"] > 20 and out["status"] == ref["status"] == 0
    assert np.array_equal(out["ids"], ref["ids"]) and np.array_equal(out["accepted"], ref["accepted"])
    assert np.array_equal(out["dx"], ref["dx"]) and np.array_equal(b.cov_download(n), Pa)
    assert b.db_size() == a.db_size()
    dx = out["dx"]
    for i, sid in enumerate(st_b.ids):
        np.testing.assert_array_equal(st_b.p[i], p0[i] + dx[sid + 3:sid + 6])
    np.testing.assert_array_equal(K, K0 + dx[sc["intr_id"]:sc["intr_id"] + 8])
    np.testing.assert_array_equal(np.array(st_b.c.intrinsics), K)          # the mirror: the view is current
    a.close(), b.close()


def test_try_update_without_variables_leaves_the_state_alone(pkg):
    fo = oracle_lib.load_front()
    sc = synth.vio_scene(F=30, M=15, noise_px=0.4, seed=3)
    n, t = sc["n_state"], sc["t"]
    st, _ = synth.scene_views(pkg, sc)
    p0 = st.p.copy()
    ctx = _filled_context(pkg, sc, fo)
    out, _, _ = ctx.camera_try_update(st, None, n, 40, 15, lines=False, t_prev_frame=t[-2], state_time=t[-1], **TRI)
    assert out["n_accepted"] > 5 and np.abs(out["dx"]).max() > 0
    np.testing.assert_array_equal(st.p, p0)
    ctx.close()


def test_camera_frame_without_update_is_the_feed(pkg):
    """update = None: tracker feed (+ line feed) only, as before the filter is initialised."""
    canvas = synth.texture_canvas(752, 480, seed=42)
    sc = synth.vio_scene(F=4, M=4)
    st, _ = synth.scene_views(pkg, sc)
    a, b = pkg.Context(pkg.default_config(752, 480)), pkg.Context(pkg.default_config(752, 480))
    for k in range(3):
        frame = synth.render_frame(canvas, 752, 480, tx=2.0 * k, ty=-1.0 * k)
        a.tracker_feed(0.1 * k, frame)
        vps = a.vanishing_points(np.array(st.c.R_ItoC).reshape(3, 3), np.array(st.c.intrinsics))
        a.line_tracker_feed(0.1 * k, vps)
        pts, lns, n_db = b.camera_frame(st, 0.1 * k, img=frame, use_lines=True, update=None)
        assert pts is None and lns is None and n_db == a.line_db_size()
    pa, ia = a.tracker_last()
    pb, ib = b.tracker_last()
    assert len(ia) > 50 and np.array_equal(ia, ib) and np.array_equal(pa, pb)
    a.close(), b.close()
