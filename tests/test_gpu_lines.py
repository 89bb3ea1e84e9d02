"""GPU parity of the line update path (a27-a29) against the oracle, through the C-ABI."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jo(pkg):
    return oracle_lib.load_jac(pkg)


def make(pkg, sc, ls, **kw):
    st, _ = synth.scene_views(pkg, sc, **kw)
    lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=ls["lines"])
    return st, lt


@pytest.mark.parametrize("calib_dt,fej_noise,offset,pol", [(False, 0.0, 0.0, 0), (True, 1e-3, 0.013, 1), (False, 2e-3, 0.02, 1)])
def test_line_jacobians_parity(ctx, pkg, jo, calib_dt, fej_noise, offset, pol):
    sc = synth.vio_scene(F=4, calib_int=True, fej_noise=fej_noise, obs_offset=offset)
    ls = synth.line_scene(sc, L=80, noise_px=0.7)
    # line observations share the point observations' time stamps: off-clone times exercise the interpolation
    ls["obs_time"] = ls["obs_time"] + np.where(ls["obs_time"] < sc["t"][-1], offset, 0.0)
    st, lt = make(pkg, sc, ls, use_pol_cov=pol, intr_ori_cov=1e-5, intr_pos_cov=2e-5, dt_state_id=14 if calib_dt else -1)
    cols = ctx.line_jacobian_columns(st, lt)
    assert (cols == jo.line_columns(st, lt)).all()
    assert len(cols) == 90 + (1 if calib_dt else 0)
    ld = 32
    rows, Hf, Hx, res = ctx.build_line_jacobians(st, lt, cols, ld)
    rows_o, Hf_o, Hx_o, res_o = jo.build_line_jacobians(st, lt, cols, ld)
    assert (rows == rows_o).all() and rows.max() == 30
    # ln_2 = l0^2 + l1 + l1 (REF quirk, LineHelper.cpp:921) can be negative: such rows are NaN in the
    # reference too (the chi2 gate then rejects the line) and must be NaN in the same places
    for a, b in ((Hf, Hf_o), (Hx, Hx_o), (res, res_o)):
        assert (np.isnan(a) == np.isnan(b)).all()
        fin = ~np.isnan(b)
        assert np.abs(a[fin] - b[fin]).max() <= 1e-9 * max(1.0, np.abs(b[fin]).max())


def test_line_jacobians_with_the_cpi_covariance_as_noise(ctx, pkg, jo):
    """est.use_imu_cov, line twin (LineHelper's noise block): parity with res_R / res_p / res_Q / res_clone on the tracks."""
    sc = synth.vio_scene(F=4, calib_int=True, obs_offset=0.013)
    ls = synth.line_scene(sc, L=40, noise_px=0.7)
    ls["obs_time"] = ls["obs_time"] + np.where(ls["obs_time"] < sc["t"][-1], 0.013, 0.0)
    rng = np.random.default_rng(4)
    nobs = len(ls["obs_time"])
    res_R = np.array([synth._exp_so3(rng.normal(0, 1e-3, 3)) @ sc["pose_fn"](t)[0] for t in ls["obs_time"]])
    res_p = np.array([sc["pose_fn"](t)[1] + rng.normal(0, 1e-3, 3) for t in ls["obs_time"]])
    A = rng.normal(0, 1.0, (nobs, 6, 6)) * np.array([2e-3] * 3 + [8e-3] * 3)[None, :, None]
    Q, ci = (A @ np.transpose(A, (0, 2, 1))).reshape(nobs, 36), rng.integers(0, len(sc["t"]), nobs).astype(np.int32)
    norms = {}
    for mode in (0, 1):
        st, _ = make(pkg, sc, ls, use_imu_cov=mode, intr_err_mlt=3.0)
        lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], line_FinG=ls["lines"], res_R=res_R, res_p=res_p,
                            res_Q=Q, res_clone=ci)
        cols = ctx.line_jacobian_columns(st, lt)
        rows, Hf, Hx, res = ctx.build_line_jacobians(st, lt, cols, 32)
        rows_o, Hf_o, Hx_o, res_o = jo.build_line_jacobians(st, lt, cols, 32)
        assert (rows == rows_o).all()
        for a, b in ((Hf, Hf_o), (Hx, Hx_o), (res, res_o)):
            assert (np.isnan(a) == np.isnan(b)).all()
            fin = ~np.isnan(b)
            assert np.abs(a[fin] - b[fin]).max() <= 1e-9 * max(1.0, np.abs(b[fin]).max())
        norms[mode] = np.nansum(np.abs(Hx_o))
    assert norms[1] != norms[0]   # the covariance does enter


def test_line_triangulation_parity(ctx, pkg, jo):
    sc = synth.vio_scene(F=4, calib_int=False, dt_clone=0.5)
    ls = synth.line_scene(sc, L=60, noise_px=0.3, depth=(4.0, 14.0))
    st, _ = synth.scene_views(pkg, sc)
    rng = np.random.default_rng(1)
    D = rng.integers(0, 4, 60)
    has = rng.integers(0, 2, 60).astype(np.uint8)
    lt = pkg.LineTracks(ls["obs_ptr"], ls["obs_time"], ls["seg_uv"], seg_uvn=ls["seg_uvn"], D=D,
                        anchor_pt=rng.normal(size=(60, 3)) * 5, has_pt=has)
    out, ok = ctx.triangulate_lines(st, lt)
    out_o, ok_o = jo.triangulate_lines(st, lt)
    assert (ok == ok_o).all() and ok.sum() > 30 and ((D > 0) & (has > 0) & (ok > 0)).sum() > 5
    assert np.abs(out - out_o).max() <= 1e-9 * max(1.0, np.abs(out_o).max())


def test_lines_update_at_configs3_size(ctx, pkg, jo, oracle):
    """BASELINE configs[3]: 150 lines seen from a 20-clone window (up to 20 observations each: ld = 40, k = 6 * 20 + the time offset),
    Jacobians -> null space -> gate -> compression -> EKFUpdate on the device against the oracle."""
    sc = synth.vio_scene(n_clones=20, F=4, calib_int=True, w=1280, h=720)
    ls = synth.line_scene(sc, L=150, M=20, noise_px=0.4, w=1280, h=720)
    st, lt = make(pkg, sc, ls, dt_state_id=14)
    cols = ctx.line_jacobian_columns(st, lt)
    assert len(cols) == 121
    ld = 40
    n = sc["n_state"]
    P = synth.spd_cov(n, seed=6) * 1e-4
    q95 = synth.q95_table()
    rows_o, Hf_o, Hx_o, res_o = jo.build_line_jacobians(st, lt, cols, ld)
    assert rows_o.max() == 40
    rc_o, P_o, dx_o, acc_o, nrows_o = oracle.msckf_update(P, rows_o, Hf_o, Hx_o, res_o, cols, 2.25, q95, res_norm_gate=0.0)
    ctx.cov_upload(P)
    ctx.build_line_jacobians_resident(st, lt, cols, ld)
    rc, dx, acc, nrows = ctx.msckf_update_resident(n, 2.25, res_norm_gate=0.0)
    Pn = ctx.cov_download(n)
    assert rc == rc_o == 0
    assert (acc == acc_o).all() and acc.sum() > 20 and nrows == nrows_o
    assert np.abs(dx - dx_o).max() <= 1e-7 * max(1.0, np.abs(dx_o).max())
    assert np.abs(Pn - P_o).max() <= 1e-8 * np.abs(P).max()


def test_lines_update_end_to_end(ctx, pkg, jo, oracle):
    """UpdaterCamera::lines_update: Jacobians -> nullspace (6) -> chi2-only gate -> compress -> EKF."""
    sc = synth.vio_scene(F=4, calib_int=True)
    ls = synth.line_scene(sc, L=80, noise_px=0.4)
    st, lt = make(pkg, sc, ls)
    cols = ctx.line_jacobian_columns(st, lt)
    ld = 32
    n = sc["n_state"]
    P = synth.spd_cov(n, seed=4) * 1e-4
    q95 = synth.q95_table()
    rows_o, Hf_o, Hx_o, res_o = jo.build_line_jacobians(st, lt, cols, ld)
    rc_o, P_o, dx_o, acc_o, nrows_o = oracle.msckf_update(P, rows_o, Hf_o, Hx_o, res_o, cols, 2.25, q95, res_norm_gate=0.0)
    ctx.cov_upload(P)
    ctx.build_line_jacobians_resident(st, lt, cols, ld)
    rc, dx, acc, nrows = ctx.msckf_update_resident(n, 2.25, res_norm_gate=0.0)
    Pn = ctx.cov_download(n)
    assert rc == rc_o == 0
    assert (acc == acc_o).all() and acc.sum() > 10 and nrows == nrows_o
    assert np.abs(dx - dx_o).max() <= 1e-7 * max(1.0, np.abs(dx_o).max())
    assert np.abs(Pn - P_o).max() <= 1e-8 * np.abs(P).max()
