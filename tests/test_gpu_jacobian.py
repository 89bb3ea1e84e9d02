"""GPU parity tests of K10 (per-feature Jacobians) through the C-ABI: HIP vs the CPU oracle on the
same seeded scenes.  fp64 with different libm (sin/cos/acos/pow): relative 1e-9."""
import os

import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jo(pkg):
    return oracle_lib.load_jac(pkg)


def _cmp(rows0, a, rows1, b, tol=1e-9):
    assert np.array_equal(rows0, rows1)
    for x, y in zip(a, b):
        scale = max(1.0, np.abs(x).max())
        assert np.max(np.abs(x - y)) <= tol * scale, np.max(np.abs(x - y))


@pytest.mark.parametrize("kw", [dict(), dict(obs_offset=0.017), dict(n_clones=20, F=150, M=20), dict(n_clones=6, F=5, M=4),
                                dict(fej_noise=1e-3)])
def test_build_jacobians_parity(pkg, ctx, jo, kw):
    sc = synth.vio_scene(**kw)
    extra = dict(use_pol_cov=1, intr_ori_cov=1e-6, intr_pos_cov=1e-6) if "obs_offset" in kw else {}
    st, tr = synth.scene_views(pkg, sc, **extra)
    cols0 = jo.columns(st, tr)
    cols1 = ctx.jacobian_columns(st, tr)
    assert np.array_equal(cols0, cols1)
    ld = 2 * kw.get("M", 15)
    r0, Hf0, Hx0, res0 = jo.build_jacobians(st, tr, cols0, ld)
    r1, Hf1, Hx1, res1 = ctx.build_jacobians(st, tr, cols1, ld)
    _cmp(r0, (Hf0, Hx0, res0), r1, (Hf1, Hx1, res1))


def test_build_jacobians_variants(pkg, ctx, jo):
    """inverse-depth representation, extrinsic + time-offset calibration columns, provided residual
    poses (use_imu_res), dropped measurements."""
    sc = synth.vio_scene(n_clones=10, F=12, M=8, obs_offset=0.011)
    sc["obs_time"] = sc["obs_time"].copy()
    sc["obs_time"][3] = sc["t"][0] - 5.0  # no bounding clones -> dropped
    nobs = len(sc["obs_time"])
    rng = np.random.default_rng(0)
    res_R = np.array([synth._exp_so3(rng.normal(0, 1e-3, 3)) @ sc["pose_fn"](t)[0] for t in sc["obs_time"]])
    res_p = np.array([sc["pose_fn"](t)[1] + rng.normal(0, 1e-3, 3) for t in sc["obs_time"]])
    st = pkg.StateView(sc["t"], sc["R"], sc["p"], sc["ids"], sc["R_ItoC"], sc["p_IinC"], sc["K8"], intrinsic_state_id=15,
                       extrinsic_state_id=sc["n_state"], dt_state_id=sc["n_state"] + 6, cam_dt=0.003, sigma_pix=1.0,
                       use_pol_cov=1, intr_ori_cov=1e-6, intr_pos_cov=1e-6, feat_rep=1)
    tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], sc["pts"], res_R=res_R, res_p=res_p)
    cols = ctx.jacobian_columns(st, tr)
    assert np.array_equal(cols, jo.columns(st, tr)) and len(cols) == 6 + 8 + 1 + 54  # clone 0 is never interpolated over
    a = jo.build_jacobians(st, tr, cols, 16)
    b = ctx.build_jacobians(st, tr, cols, 16)
    _cmp(a[0], a[1:], b[0], b[1:])
    # feature 0: one observation without bounding clones, and the newest one lands 3 ms (cam_dt) past the
    # newest clone (REF: State.cpp:852-855) -> both dropped
    assert a[0][0] == 2 * (sc["obs_ptr"][1] - 2)


def _cpi_cov(rng, nobs, n_clones):
    """a CPI covariance (SPD 6 x 6, gamma / alpha blocks) and the clone it hangs on, per observation"""
    A = rng.normal(0, 1.0, (nobs, 6, 6)) * np.array([2e-3] * 3 + [8e-3] * 3)[None, :, None]
    return (A @ np.transpose(A, (0, 2, 1))).reshape(nobs, 36), rng.integers(0, n_clones, nobs).astype(np.int32)


def test_jacobians_with_the_cpi_covariance_as_noise(pkg, ctx, jo):
    """est.use_imu_cov (REF CamHelper.cpp:217-224): R += H_ Q H_^T * mlt for poses between clones."""
    sc = synth.vio_scene(n_clones=10, F=12, M=8, obs_offset=0.011)
    rng = np.random.default_rng(0)
    res_R = np.array([synth._exp_so3(rng.normal(0, 1e-3, 3)) @ sc["pose_fn"](t)[0] for t in sc["obs_time"]])
    res_p = np.array([sc["pose_fn"](t)[1] + rng.normal(0, 1e-3, 3) for t in sc["obs_time"]])
    Q, ci = _cpi_cov(rng, len(sc["obs_time"]), len(sc["t"]))
    out = {}
    for mode in (0, 1):
        st = pkg.StateView(sc["t"], sc["R"], sc["p"], sc["ids"], sc["R_ItoC"], sc["p_IinC"], sc["K8"], intrinsic_state_id=15,
                           extrinsic_state_id=sc["n_state"], dt_state_id=sc["n_state"] + 6, cam_dt=0.003, sigma_pix=1.0,
                           use_imu_cov=mode, intr_err_mlt=3.0)
        tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], sc["pts"], res_R=res_R, res_p=res_p, res_Q=Q, res_clone=ci)
        cols = ctx.jacobian_columns(st, tr)
        a = jo.build_jacobians(st, tr, cols, 16)
        b = ctx.build_jacobians(st, tr, cols, 16)
        assert np.array_equal(a[0], b[0])
        for x, y in zip(a[1:], b[1:]):   # (a strongly correlated R makes the reference's whitening NaN: same places on both sides)
            assert np.array_equal(np.isnan(x), np.isnan(y))
            fin = ~np.isnan(x)
            assert np.abs(x[fin] - y[fin]).max() <= 1e-9 * max(1.0, np.abs(x[fin]).max())
        out[mode] = a
    assert not np.array_equal(np.nan_to_num(out[1][2]), np.nan_to_num(out[0][2]))   # the covariance does enter


def test_update_from_tracks_end_to_end(pkg, ctx, jo, oracle):
    """Jacobians -> nullspace -> gate -> compress -> EKF entirely on the device vs the oracle chain."""
    # the reference gates on the norm of the WHITENED residual (< 3, UpdaterCamera.cpp:242): 27 rows of
    # 1 px noise / sigma 1.5 already give ~3.5, so long tracks only pass with sub-pixel noise
    sc = synth.vio_scene(noise_px=0.4)
    st, tr = synth.scene_views(pkg, sc)
    cols = ctx.jacobian_columns(st, tr)
    n = sc["n_state"]
    P = synth.spd_cov(n)
    rows, Hf, Hx, res = jo.build_jacobians(st, tr, cols, 30)
    rc0, P0, dx0, acc0, nr0 = oracle.msckf_update(P, rows, Hf, Hx, res, cols, 1.0, synth.q95_table())
    ctx.cov_upload(P)
    ctx.build_jacobians_resident(st, tr, cols, 30)
    rc1, dx1, acc1, nr1 = ctx.msckf_update_resident(n, 1.0)
    assert rc0 == 0 and rc1 == 0
    assert np.array_equal(acc0, acc1) and nr0 == nr1 and acc1.sum() > 30
    assert np.max(np.abs(dx1 - dx0)) <= 1e-7 * np.max(np.abs(dx0))
    P1 = ctx.cov_download(n)
    assert np.max(np.abs(P1 - P0)) <= 1e-7 * np.max(np.abs(P0))
    # the device-built batch is single use
    with pytest.raises(pkg.PlvError):
        ctx.msckf_update_resident(n, 1.0)


def test_triangulate_batch_parity(pkg, ctx, jo):
    """a18/a19: camera poses from the estimate polynomial, linear triangulation, LM refinement,
    reprojection error — thread per feature on the device vs the oracle."""
    fo = oracle_lib.load_front()
    for kw, opt in ((dict(noise_px=0.3), dict(max_cond=1e7, max_dist=150.0, max_baseline=2000.0)),
                    (dict(noise_px=0.3, obs_offset=0.013), dict(max_cond=1e7, max_dist=150.0, max_baseline=2000.0)),
                    (dict(noise_px=1.0), dict())):
        sc = synth.vio_scene(n_clones=15, F=60, M=15, **kw)
        uvn = fo.undistort(sc["K8"], sc["obs_uv"])
        st = pkg.StateView(sc["t"], sc["R"], sc["p"], sc["ids"], sc["R_ItoC"], sc["p_IinC"], sc["K8"], intrinsic_state_id=15)
        tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], np.zeros((60, 3)), obs_uvn=uvn)
        p0, ok0, e0 = jo.triangulate_batch(st, tr, **opt)
        p1, ok1, e1 = ctx.triangulate(st, tr, **opt)
        assert np.array_equal(ok0, ok1)
        good = ok0.astype(bool)
        if "max_cond" in opt:
            assert good.sum() >= 40
            rel = np.linalg.norm(p1[good] - sc["pts"][good], axis=1) / np.linalg.norm(sc["pts"][good] - sc["p"][-1], axis=1)
            assert np.median(rel) < 0.1
        assert np.max(np.abs(p1[good] - p0[good])) <= 1e-6 * max(1.0, np.abs(p0[good]).max()) if good.any() else True
        assert np.allclose(e1[good], e0[good], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,n_clones,offset", [(17, 18, 0.0), (21, 21, 0.004), (21, 22, -0.004), (23, 24, 0.0), (11, 11, -0.004)])
def test_triangulate_long_tracks_and_window_edges(pkg, ctx, jo, M, n_clones, offset):
    """Tracks longer than 16 observations take the refinement's one-candidate-at-a-time path (the four-candidate passes are for up to
    16), and observations offset from the clone times reach the interpolation windows at both ends of the clone list: position to
    1e-9 against the oracle, the reprojection error with it."""
    fo = oracle_lib.load_front()
    sc = synth.vio_scene(n_clones=n_clones, F=60, M=M, noise_px=0.3, obs_offset=offset)
    uvn = fo.undistort(sc["K8"], sc["obs_uv"])
    st = pkg.StateView(sc["t"], sc["R"], sc["p"], sc["ids"], sc["R_ItoC"], sc["p_IinC"], sc["K8"], intrinsic_state_id=15)
    tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], np.zeros((60, 3)), obs_uvn=uvn)
    opt = dict(max_cond=1e7, max_dist=150.0, max_baseline=2000.0)
    p0, ok0, e0 = jo.triangulate_batch(st, tr, **opt)
    p1, ok1, e1 = ctx.triangulate(st, tr, **opt)
    assert np.array_equal(ok0, ok1) and ok0.sum() >= 50
    good = ok0.astype(bool)
    assert np.abs(p1[good] - p0[good]).max() < 1e-9 and np.allclose(e1[good], e0[good], rtol=1e-7, atol=1e-9)


def test_triangulate_with_two_window_poses_at_one_instant(pkg, ctx, jo):
    """A batch captured from a replay whose camera was stamped on IMU sample instants (round 4): the newest clone and the IMU
    pose that closes the window are 9e-16 s apart, and the interpolation polynomial through such a pair is undefined — the reference's
    as much as this one's.  Tracks that do not reach the pair agree to rounding; the others are triangulated by both implementations
    (same verdicts), a few centimetres apart.  The synthetic drives no longer produce such windows; the case stays as a record of why."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "triangulate_window_edge.npz"))
    assert d["t"][-1] - d["t"][-2] < 1e-12 and len(d["t"]) == 22
    st = pkg.StateView(d["t"], d["R"], d["p"], d["ids"], d["R_ItoC"], d["p_IinC"], d["K8"], clone_R_fej=d["Rf"], clone_p_fej=d["pf"], cam_dt=float(d["cam_dt"]),
                       dt_exp=float(d["dt_exp"]))
    tr = pkg.Tracks(d["ptr"], d["ot"], d["uv"], np.zeros((len(d["ptr"]) - 1, 3)), obs_uvn=d["uvn"])
    kw = dict(zip(("min_dist", "max_dist", "max_cond", "max_baseline"), (float(x) for x in d["kw"])))
    p0, ok0, _ = jo.triangulate_batch(st, tr, **kw)
    p1, ok1, _ = ctx.triangulate(st, tr, **kw)
    assert np.array_equal(ok0, ok1)
    good, nobs = ok0.astype(bool), np.diff(d["ptr"])
    short = good & (nobs < 21)
    assert short.sum() >= 5 and np.abs(p1[short] - p0[short]).max() < 1e-8
    assert np.abs(p1[good] - p0[good]).max() < 0.2


def test_cpi_poses_parity_and_use(pkg):
    """a19 use_imu_res: plv_cpi_poses against the oracle, then the poses as the residual poses of the Jacobian build."""
    from test_oracle_cpi import _setup
    jo = oracle_lib.load_jac(pkg)
    sc, cp, st, tab = _setup(pkg, obs_offset=0.0125)
    ctx = pkg.Context(pkg.default_config(752, 480))
    rng = np.random.default_rng(2)
    t = cp["t"]
    last_of_run = t[:-1][np.diff(cp["clone_t"]) != 0]
    tq = np.concatenate([t[::2], t[:-1] + rng.uniform(0.0002, 0.0048, len(t) - 1), [t[0] - 1.0, t[-1] + 1.0, last_of_run[1] + 1e-4]])
    rng.shuffle(tq)
    R, p, ok = ctx.cpi_poses(st, tab, tq)
    Ro, po, oko = jo.cpi_poses(st, tab, tq)
    assert np.array_equal(ok, oko) and 0 < (ok == 0).sum() < 40
    assert np.abs(R - Ro).max() < 1e-12 and np.abs(p - po).max() < 1e-12
    # a shrunken window: records of marginalised clones drop out identically
    st2 = pkg.StateView(sc["t"][2:], sc["R"][2:], sc["p"][2:], sc["ids"][2:], sc["R_ItoC"], sc["p_IinC"], sc["K8"])
    R2, p2, ok2 = ctx.cpi_poses(st2, tab, tq)
    Ro2, po2, oko2 = jo.cpi_poses(st2, tab, tq)
    assert np.array_equal(ok2, oko2) and ok2.sum() < ok.sum()
    assert np.abs(R2 - Ro2).max() < 1e-12 and np.abs(p2 - po2).max() < 1e-12
    # the observation times of the scene (clone time + 12.5 ms) through the CPI table -> residual poses of a19/a20
    Rq, pq, okq = ctx.cpi_poses(st, tab, sc["obs_time"])
    assert okq.all()
    tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], sc["pts"], res_R=Rq, res_p=pq)
    cols = ctx.jacobian_columns(st, tr)
    rows, Hf, Hx, res = ctx.build_jacobians(st, tr, cols, 30)
    rows_o, Hf_o, Hx_o, res_o = jo.build_jacobians(st, tr, cols, 30)
    assert np.array_equal(rows, rows_o)
    assert np.abs(res - res_o).max() < 1e-9 and np.abs(Hx - Hx_o).max() < 1e-9 * max(1.0, np.abs(Hx_o).max())
    # exact IMU poses instead of the cubic through the clones: the residual is pixel noise again
    tr_poly = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], sc["pts"])
    _, _, _, res_poly = ctx.build_jacobians(st, tr_poly, cols, 30)
    assert np.abs(res).max() < 4.0 / 1.5 * 1.5 and np.abs(res).max() <= np.abs(res_poly).max() + 1e-9
    ctx.close()


def test_fused_build_project_paths_agree(pkg, oracle):
    """plv_build_jacobians_resident builds AND projects in one launch and lets the update's covariance gathers ride on it; the
    gathers are only reused while nothing has touched the covariance since.  Three routes, one result."""
    jo = oracle_lib.load_jac(pkg)
    sc = synth.vio_scene(F=50, M=15, noise_px=0.4, seed=21)
    st, tr = synth.scene_views(pkg, sc)
    n = sc["n_state"]
    P = synth.spd_cov(n, seed=4) * 1e-4
    ctx = pkg.Context(pkg.default_config(752, 480))
    cols = ctx.jacobian_columns(st, tr)
    s2 = st.c.sigma_pix ** 2
    # (a) host systems -> the general update entry point (separate nullspace launch)
    rows, Hf, Hx, res = ctx.build_jacobians(st, tr, cols, 30)
    rc_a, P_a, dx_a, acc_a, nr_a = ctx.msckf_update(P, rows, Hf, Hx, res, cols, s2)
    rc_o, P_o, dx_o, acc_o, _ = oracle.msckf_update(P, *jo.build_jacobians(st, tr, cols, 30), cols, s2, synth.q95_table())
    assert rc_a == rc_o == 0 and np.array_equal(acc_a, acc_o) and acc_a.sum() > 20
    # (b) resident, gathers reused
    ctx.cov_upload(P)
    ctx.build_jacobians_resident(st, tr, cols, 30)
    rc_b, dx_b, acc_b, nr_b = ctx.msckf_update_resident(n, s2)
    P_b = ctx.cov_download(n)
    # (c) resident, covariance replaced between the build and the update: the gathers must be redone
    ctx.cov_upload(P * 0.5)
    ctx.build_jacobians_resident(st, tr, cols, 30)
    ctx.cov_upload(P)
    rc_c, dx_c, acc_c, nr_c = ctx.msckf_update_resident(n, s2)
    P_c = ctx.cov_download(n)
    for rc, dx, acc, nr, Pn in ((rc_b, dx_b, acc_b, nr_b, P_b), (rc_c, dx_c, acc_c, nr_c, P_c)):
        assert rc == 0 and np.array_equal(acc, acc_a) and nr == nr_a
        # (the fused launch projects with Householder reflections, route (a) with the reference's Givens order: same null space,
        #  another orthonormal basis of it)
        assert np.abs(dx - dx_a).max() <= 1e-9 * max(1.0, np.abs(dx_a).max()) and np.abs(Pn - P_a).max() <= 1e-9 * np.abs(P_a).max()
    assert np.abs(P_a - P_o).max() <= 1e-8 * np.abs(P_o).max() and np.abs(P_b - P_c).max() <= 1e-13 * np.abs(P_b).max()
    ctx.close()
