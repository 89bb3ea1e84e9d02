"""plv_state_boxplus (x <- x [+] dx for the whole state in one call) against the per-variable updates of the reference's types
(REF: StateHelper.cpp:156-160; ov_type::Vec / JPLQuat / PoseJPL::update), restated here with numpy."""
import ctypes as C
import importlib

import numpy as np
import pytest

import __graft_entry__ as ge


def quat_left_update_ref(q, dth):     # JPLQuat.h:62-73, quat_ops.h:152-157,232-252
    dq = np.concatenate([0.5 * dth, [1.0]])
    dq = dq / np.linalg.norm(dq)
    a, b, v, w = dq[:3], dq[3], q[:3], q[3]
    r = np.concatenate([b * v - np.cross(a, v) + a * w, [-(a @ v) + b * w]])
    if r[3] < 0:
        r = -r
    return r / np.linalg.norm(r)


def rot_ref(q):                        # quat_2_Rot
    x, y, z, w = q
    sk = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return (2 * w * w - 1) * np.eye(3) - 2 * w * sk + 2 * np.outer(q[:3], q[:3])


def test_boxplus_matches_the_per_variable_updates():
    pkg = ge.load_pkg()
    rng = np.random.default_rng(3)
    n_dx = 40
    dx = 0.05 * rng.standard_normal(n_dx)
    q = rng.standard_normal((3, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    q[1] *= -1 if q[1, 3] > 0 else 1          # one with w < 0 on the way in
    q0 = q.copy()
    p, p0 = None, None
    p = rng.standard_normal((3, 3))
    p0 = p.copy()
    K = rng.standard_normal(8)
    K0 = K.copy()
    R = np.zeros((3, 9))
    mirR, mirK = np.zeros(9), np.zeros(8)
    ent = [("quat", 0, q[0], R[0], None), ("vec", 3, p[0], None, None), ("quat", 15, q[1], R[1], mirR), ("vec", 18, p[1], None, None),
           ("vec", 21, K, None, mirK), ("quat", 34, q[2], None, None), ("vec", 37, p[2], None, None)]
    plus = pkg.BoxPlus(ent)
    plus.apply(dx)
    for i, (qi, pi) in enumerate(((0, 3), (15, 18), (34, 37))):
        want = quat_left_update_ref(q0[i], dx[qi:qi + 3])
        np.testing.assert_allclose(q[i], want, rtol=0, atol=2e-16)
        assert q[i, 3] >= 0 and abs(np.linalg.norm(q[i]) - 1) < 1e-15
        np.testing.assert_array_equal(p[i], p0[i] + dx[pi:pi + 3])
    np.testing.assert_allclose(R[0].reshape(3, 3), rot_ref(q[0]), rtol=0, atol=5e-16)
    np.testing.assert_array_equal(mirR, R[1])
    assert not R[2].any()                                  # no output asked for
    np.testing.assert_array_equal(K, K0 + dx[21:29])
    np.testing.assert_array_equal(mirK, K)


def test_boxplus_rejects_variables_outside_dx():
    pkg = ge.load_pkg()
    v = np.zeros(3)
    with pytest.raises(pkg.PlvError):
        pkg.BoxPlus([("vec", 8, v, None, None)]).apply(np.zeros(10))
    with pytest.raises(pkg.PlvError):
        pkg.BoxPlus([("quat", 8, np.array([0, 0, 0, 1.0]), None, None)]).apply(np.zeros(10))
    pkg.BoxPlus([("vec", 7, v, None, None)]).apply(np.ones(10))
    np.testing.assert_array_equal(v, 1.0)


def test_state_apply_keeps_view_and_poses_in_step():
    """State.apply through the prepared call: the clone Pose objects, the window arrays and the state view's calibration fields move
    together and agree with the per-variable Python updates."""
    pkg = ge.load_pkg()
    system = importlib.import_module(pkg.__name__ + ".system")
    options = importlib.import_module(pkg.__name__ + ".options")

    class FakeCtx:
        def cov_upload(self, P): self.n = len(P)
        def cov_clone(self, *a): pass
        def set_camera_intrinsics(self, K): self.K = np.array(K)
    import os
    op = options.load_options(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_sample", "config.yaml"))
    op.est.cam.do_calib_ext = op.est.cam.do_calib_int = op.est.cam.do_calib_dt = True
    st = system.State(op, FakeCtx())
    rng = np.random.default_rng(5)
    for k in range(5):
        st.time = 1.0 + 0.1 * k
        x = np.frombuffer(st.imu, dtype=np.float64)
        qq = rng.standard_normal(4)
        x[0:4] = qq / np.linalg.norm(qq) * (1 if qq[3] > 0 else -1)
        x[4:7] = rng.standard_normal(3)
        x[16:20], x[20:23] = x[0:4], x[4:7]
        st.augment_clone()
    sv = st.view()
    before = {t: (c.q.copy(), c.p.copy()) for t, c in st.clones.items()}
    ext_q, ext_p, K, dt = st.cam_ext.q.copy(), st.cam_ext.p.copy(), st.cam_intr.v.copy(), st.cam_dt.v.copy()
    dx = 0.01 * rng.standard_normal(st.n)
    st.apply(dx)
    sv2 = st.view()
    assert sv2 is sv
    for i, t in enumerate(sorted(st.clones)):
        c = st.clones[t]
        np.testing.assert_allclose(c.q, quat_left_update_ref(before[t][0], dx[c.id:c.id + 3]), rtol=0, atol=2e-16)
        np.testing.assert_array_equal(c.p, before[t][1] + dx[c.id + 3:c.id + 6])
        np.testing.assert_allclose(sv.R[i].reshape(3, 3), rot_ref(c.q), rtol=0, atol=5e-16)
        np.testing.assert_array_equal(c.Rot().ravel(), sv.R[i])
        np.testing.assert_array_equal(sv.p[i], c.p)
    e = st.cam_ext
    np.testing.assert_allclose(e.q, quat_left_update_ref(ext_q, dx[e.id:e.id + 3]), rtol=0, atol=2e-16)
    np.testing.assert_array_equal(e.p, ext_p + dx[e.id + 3:e.id + 6])
    np.testing.assert_allclose(np.array(sv.c.R_ItoC).reshape(3, 3), rot_ref(e.q), rtol=0, atol=5e-16)
    np.testing.assert_array_equal(np.array(sv.c.R_ItoC), e.Rot().ravel())
    np.testing.assert_array_equal(np.array(sv.c.p_IinC), e.p)
    np.testing.assert_array_equal(np.array(sv.c.intrinsics), K + dx[st.cam_intr.id:st.cam_intr.id + 8])
    assert sv.c.cam_dt == dt[0] + dx[st.cam_dt.id]
    np.testing.assert_array_equal(st.ctx.K, st.cam_intr.v)
